#!/usr/bin/env python3
"""
tests/golden/make_golden_2d.py -- pin the d > 1 branch (jets, Laplacian, H u) to THE REFERENCE ITSELF.

The reference's 2D class (Gross-Pitaevskii/src/gross_pitaevskii_2D_minimal.py) has the broadcasting quirk Q1: for N points
its residual is [N,N] (V is [N], u is [N,1]; lines 170-182).  Called with ONE point per call the broadcast is inert, and
`pde_loss` then returns the intended maths of that point:
    lambda_pde = (u_x^2 + u_y^2 + V u^2 + g u^4) / u^2                      (line 179: the energy-functional lambda)
    residual   = -(u_xx + u_yy) + V u + g |u^2| u - lambda_pde u            (line 182)
so  H u := residual + lambda_pde u  is the reference's own value of  -laplacian(u) + V u + g u^3  at that point, and
lambda_pde pins u_x^2 + u_y^2.  This script imports that class (read-only, from /root/reference), seeds it, evaluates K points
one at a time and stores inputs + outputs; it also stores u_x, u_y, u_xx, u_yy obtained from the imported model with the
autograd calls of lines 170-175.  Runs only in the build container; the .npz is committed.

Usage:  MPLBACKEND=Agg python tests/golden/make_golden_2d.py
"""
import importlib.util
import os

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch
from torch.autograd import grad

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fixture(g2d, tag, layers, g, seed, K):
    torch.manual_seed(seed)
    model = g2d.GrossPitaevskiiPINN(layers, g=g)
    flat0 = np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, np.pi, (K, 2)).astype(np.float32)          # the reference's disk lives in [0, pi]^2
    rows = []
    for k in range(K):
        xi = torch.tensor(x[k:k + 1], requires_grad=True)
        u = model.forward(xi)
        pde_loss, resid, lam = model.pde_loss(xi, u)                 # N = 1: quirk Q1 inert
        assert tuple(resid.shape) == (1, 1)
        g1 = grad(u, xi, grad_outputs=torch.ones_like(u), create_graph=True)[0]
        u_x, u_y = g1[:, 0], g1[:, 1]
        u_xx = grad(u_x, xi, grad_outputs=torch.ones_like(u_x), create_graph=True)[0][:, 0]
        u_yy = grad(u_y, xi, grad_outputs=torch.ones_like(u_y), create_graph=True)[0][:, 1]
        V = model.compute_potential(xi)
        rows.append([float(u), float(resid), float(lam), float(u_x), float(u_y), float(u_xx), float(u_yy), float(V),
                     float(pde_loss)])
    xa = torch.tensor(x, requires_grad=True)                        # riesz_loss on all K points in one call (lines 115-146)
    riesz_all = float(model.riesz_loss(model.forward(xa), xa))
    r = np.array(rows, dtype=np.float64)
    fx = dict(layers=np.array(layers), g=float(g), seed=seed, x=x, flat0=flat0, u=r[:, 0], residual=r[:, 1], lam=r[:, 2],
              u_x=r[:, 3], u_y=r[:, 4], u_xx=r[:, 5], u_yy=r[:, 6], V=r[:, 7], pde_loss=r[:, 8], riesz_all=riesz_all)
    np.savez_compressed(os.path.join(OUT, f"fx_2d_ref_points_{tag}.npz"), **fx)
    print("wrote 2d", tag, "max|u|", np.abs(r[:, 0]).max(), "max|lap|", np.abs(r[:, 5] + r[:, 6]).max())


if __name__ == "__main__":
    g2d = load_module("ref_g2d_minimal", os.path.join(REF, "Gross-Pitaevskii/src/gross_pitaevskii_2D_minimal.py"))
    fixture(g2d, "64x4_g500", [2, 64, 64, 64, 64, 1], 500.0, 0, 96)
    fixture(g2d, "100x3_g100", [2, 100, 100, 100, 1], 100.0, 1, 48)       # the reference's own architecture (line 331)
    fixture(g2d, "128x5_g500", [2, 128, 128, 128, 128, 128, 1], 500.0, 2, 48)
