#!/usr/bin/env python3
"""
tests/golden/make_golden_2d_class.py -- golden vectors for the 2D classes' LOSS (not only their jets: make_golden_2d.py), produced by
importing the reference's own script (read-only):
  /root/reference/Gross-Pitaevskii/src/gross_pitaevskii_2D_minimal.py   (class :12-222, prepare_training_data :225-261, initialize_weights :264-275)
(src/gross_pitaevskii_2D.py holds the same class -- boundary_loss :83-109, riesz_loss :112-151, pde_loss :154-213, loss :215-242 -- but imports
pyDOE, which this image does not have: it cannot be imported here, so its polar prepare_training_data :277-295 is restated from the text only.)
The class has the broadcasting quirk Q1 ([N] x [N,1] -> [N,N] for N > 1 collocation points), so every collocation call here hands it ONE
point: model.total_loss(x_k, x_bc, u_bc) = 10 mean(u(x_bc)^2) + riesz_loss + pde_loss with the energy-functional lambda (:192), whose branch of
the gradient does not vanish, and the two regularisers (:201,:204) -- value, pieces and the parameter gradient from loss.backward().
Also: the script's seeded training set and seeded initial weights.  Arrays only; runs only in the build container.

Usage:  MPLBACKEND=Agg python tests/golden/make_golden_2d_class.py
"""
import importlib.util
import os

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch

REF = "/root/reference/Gross-Pitaevskii/src"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def flat_params(model):
    return np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


def class_fixture(g2d, tag, layers, g, seed, K, N_u):
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = g2d.GrossPitaevskiiPINN(layers, g=g)
    model.apply(g2d.initialize_weights)
    flat0 = flat_params(model)
    X_f, X_u, u_train = g2d.prepare_training_data(N_u, 2 * K)          # draws in the square, kept inside the disk: the first K of them; N_u on the circle
    assert X_f.shape[0] >= K
    x = X_f[:K].astype(np.float32)
    xb = torch.tensor(X_u, dtype=torch.float32)
    ub = torch.tensor(u_train, dtype=torch.float32)
    rows, grads = [], []
    for k in range(K):
        xi = torch.tensor(x[k:k + 1], requires_grad=True)
        model.zero_grad()
        total = model.total_loss(xi, xb, ub)                            # :201-222 (N = 1 collocation point: quirk Q1 inert)
        total.backward()
        grads.append(np.concatenate([q.grad.detach().numpy().ravel() for q in model.parameters()]).astype(np.float32))
        u = model.forward(xi)
        pde, resid, lam = model.pde_loss(xi, u)
        assert tuple(resid.shape) == (1, 1)
        riesz = model.riesz_loss(model.forward(xi), xi)
        bc = model.boundary_loss(xb, ub)
        rows.append([float(total), float(pde), float(resid), float(lam), float(riesz), float(bc), float(u)])
    r = np.array(rows, dtype=np.float64)
    fx = dict(layers=np.array(layers), g=float(g), seed=seed, flat0=flat0, x=x, x_bc=X_u.astype(np.float32), total=r[:, 0], pde_loss=r[:, 1],
              residual=r[:, 2], lam=r[:, 3], riesz=r[:, 4], bc_loss=r[:, 5], u=r[:, 6], grad=np.stack(grads))
    np.savez_compressed(os.path.join(OUT, f"fx_2d_class_{tag}.npz"), **fx)
    print("wrote 2d class", tag, "total", r[:3, 0], "lam", r[:3, 3], "max|grad|", np.abs(fx["grad"]).max(), flush=True)


def data_fixture(g2d):
    np.random.seed(7)
    Xf_s, Xu_s, u_s = g2d.prepare_training_data(12, 40)                 # square draws, kept inside the disk
    torch.manual_seed(3)
    m = g2d.GrossPitaevskiiPINN([2, 16, 16, 1])
    m.apply(g2d.initialize_weights)
    np.savez_compressed(os.path.join(OUT, "fx_2d_class_data.npz"), seed=7, N_u=12, N_f=40,
                        square_X_f=Xf_s, square_X_u=Xu_s, square_u=u_s, init_seed=3, init_layers=np.array([2, 16, 16, 1]),
                        init_flat=flat_params(m))
    print("wrote 2d class data", Xf_s.shape, flush=True)


if __name__ == "__main__":
    g2d = load_module("ref_g2d_minimal2", os.path.join(REF, "gross_pitaevskii_2D_minimal.py"))
    class_fixture(g2d, "32x2_g100", [2, 32, 32, 1], 100.0, 0, 24, 16)
    class_fixture(g2d, "64x4_g500", [2, 64, 64, 64, 64, 1], 500.0, 1, 12, 24)          # the north-star architecture
    class_fixture(g2d, "100x3_g100", [2, 100, 100, 100, 1], 100.0, 2, 8, 20)          # the script's own architecture (:377)
    data_fixture(g2d)
