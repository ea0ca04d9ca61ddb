#!/usr/bin/env python3
"""
tests/golden/make_golden_box2gauss.py -- golden vectors for the RESIDUAL-BLOCK network flavour (row f3), produced by importing
/root/reference/Gross-Pitaevskii/src/final/refine/box_to_gaussian_pinn_simulation.py (read-only): its GrossPitaevskiiPINN
(use_residual=True: Linear + ShiftedTanh, ResidualBlocks tanh(lin2(tanh(lin1 x)) + x), Linear; :52-130), box sine base, Gaussian
potential exp(-(x - 0.5)^2), the epoch-0 body of its train_gpe_model (:352-380).  Runs only in the build container.

Usage:  MPLBACKEND=Agg python tests/golden/make_golden_box2gauss.py
"""
import importlib.util
import os

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def flat_params(model):
    return np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


def fixture(b2g, tag, layers, N, seed, mode, gamma, p, perturb_const, ub=1.0):
    torch.manual_seed(seed)
    lb = 0.0
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = b2g.GrossPitaevskiiPINN(layers, mode=mode, gamma=gamma, L=ub, use_residual=True)
    model.apply(lambda m: b2g.advanced_initialization(m, mode))
    keys = list(model.state_dict().keys())
    flat0 = flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)
    u_nn = model.forward(X_tensor)
    normal_const = torch.max(u_nn).detach().clone()
    u_pred = perturb_const * (u_nn / normal_const)
    u = model.get_complete_solution(X_tensor, u_pred)
    u_x = torch.autograd.grad(u, X_tensor, torch.ones_like(u), create_graph=True, retain_graph=True)[0]
    u_xx = torch.autograd.grad(u_x, X_tensor, torch.ones_like(u_x), create_graph=True, retain_graph=True)[0]
    pde_loss, lam = model.pde_loss(X_tensor, u_pred, gamma, p, "gaussian")
    bl = model.boundary_loss(bpts, bvals)
    nl = model.normalization_loss(model.get_complete_solution(X_tensor, u_pred), dx)
    total = pde_loss + 10.0 * bl + 20.0 * nl
    model.zero_grad()
    total.backward()
    grad0 = np.concatenate([q.grad.detach().numpy().ravel() for q in model.parameters()]).astype(np.float32)
    fx = dict(layers=np.array(layers), N=N, seed=seed, mode=mode, gamma=gamma, p=p, perturb_const=perturb_const,
              normal_const=float(normal_const), dx=dx, lb=lb, ub=ub, flat0=flat0, x=X.astype(np.float32),
              forward_out=u_nn.detach().numpy(), u=u.detach().numpy(), u_x=u_x.detach().numpy(), u_xx=u_xx.detach().numpy(),
              V=model.compute_potential(X_tensor).detach().numpy(), lam=float(lam), pde_loss=float(pde_loss), bc_loss=float(bl),
              norm_loss=float(nl), total=float(total), grad0=grad0, state_dict_keys=np.array(keys))
    np.savez_compressed(os.path.join(OUT, f"fx_box2gauss_{tag}.npz"), **fx)
    print("wrote box2gauss", tag, "loss0", fx['total'], "lam0", fx['lam'], keys[:4], len(flat0))


if __name__ == "__main__":
    b2g = load_module("ref_refine_b2g", os.path.join(REF, "Gross-Pitaevskii/src/final/refine/box_to_gaussian_pinn_simulation.py"))
    fixture(b2g, "m0_g0", [1, 64, 64, 64, 1], 400, 0, 0, 0.0, 3, 0.01)
    fixture(b2g, "m1_g5_p4", [1, 32, 32, 32, 32, 1], 300, 3, 1, 5.0, 4, 0.01)
