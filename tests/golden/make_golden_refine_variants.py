#!/usr/bin/env python3
"""
tests/golden/make_golden_refine_variants.py -- golden vectors for the two remaining flavours of the refine/ harmonic script family,
produced by importing the reference (read-only); runs only in the build container.

  fx_refine_neg_*   ATTRACTIVE interaction, gamma < 0: the class of refine/harmonic_pinn_simulation_negative_interaction_strength.py
                    (same residual as the main script, :146-196; its driver walks gamma downwards from 0, :286) -- same fixture layout
                    as the other fx_refine_* files (op-level tensors, gradient, 25-epoch trace with the reference's optimiser objects).
  fx_vanilla_*      use_perturbation=False of refine/harmonic_pinn_simulation.py (:152-155, :205-208): the network output IS the
                    wavefunction (no Hermite base) in pde_loss and boundary_loss; normalisation of that same prediction.

Usage:  MPLBACKEND=Agg python tests/golden/make_golden_refine_variants.py
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg          # its generators; the __main__ blocks do not run on import

REF = mg.REF
OUT = mg.OUT


def vanilla_fixture(refine, tag, layers, N, seed, mode, gamma, p, perturb_const):
    torch.manual_seed(seed)
    lb, ub = -10.0, 10.0
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    dx = X[1, 0] - X[0, 0]
    model = refine.GrossPitaevskiiPINN(layers, mode=mode, gamma=gamma, use_perturbation=False)
    model.apply(lambda m: refine.advanced_initialization(m, mode))
    flat0 = mg.flat_params(model)
    X_tensor = torch.tensor(X, dtype=torch.float32, requires_grad=True)
    bpts = torch.tensor([[lb], [ub]], dtype=torch.float32)
    bvals = torch.zeros((2, 1), dtype=torch.float32)
    u_nn = model.forward(X_tensor)
    normal_const = torch.max(u_nn).detach().clone()
    u_pred = perturb_const * (u_nn / normal_const)
    u = u_pred                                                   # :155  u = predictions
    u_x = torch.autograd.grad(u, X_tensor, torch.ones_like(u), create_graph=True, retain_graph=True)[0]
    u_xx = torch.autograd.grad(u_x, X_tensor, torch.ones_like(u_x), create_graph=True, retain_graph=True)[0]
    V = model.compute_potential(X_tensor, "harmonic")
    pde_loss, lam = model.pde_loss(X_tensor, u_pred, gamma, p, "harmonic")
    bl = model.boundary_loss(bpts, bvals)                         # :208  raw network output at the boundary
    nl = model.normalization_loss(u_pred, dx)                     # the prediction itself (the class method takes u explicitly)
    total = pde_loss + 10.0 * bl + 20.0 * nl
    model.zero_grad()
    total.backward()
    grad0 = mg.flat_grads(model)
    with torch.no_grad():
        resid = (-u_xx + V * u + gamma * u ** p) - lam * u
    fx = dict(layers=np.array(layers), N=N, seed=seed, mode=mode, gamma=gamma, p=p, perturb_const=perturb_const,
              normal_const=float(normal_const), dx=dx, lb=lb, ub=ub, flat0=flat0, x=X.astype(np.float32),
              nn_out=u_nn.detach().numpy(), u=u.detach().numpy(), u_x=u_x.detach().numpy(), u_xx=u_xx.detach().numpy(),
              V=V.detach().numpy(), lam=float(lam), residual=resid.numpy(), pde_loss=float(pde_loss), bc_loss=float(bl),
              norm_loss=float(nl), total=float(total), grad0=grad0)
    np.savez_compressed(os.path.join(OUT, f"fx_vanilla_{tag}.npz"), **fx)
    print("wrote vanilla", tag, "loss0", fx["total"], "lam0", fx["lam"])


if __name__ == "__main__":
    base = os.path.join(REF, "Gross-Pitaevskii/src/final/refine")
    neg = mg.load_module("ref_refine_negative", os.path.join(base, "harmonic_pinn_simulation_negative_interaction_strength.py"))
    mg.refine_fixture(neg, "neg_m0_gm8_64x3", [1, 64, 64, 64, 1], 400, 4, 0, -8.0, 3, 0.01, 25)
    mg.refine_fixture(neg, "neg_m1_gm2_32x4", [1, 32, 32, 32, 32, 1], 300, 5, 1, -2.0, 3, 0.01, 25)
    refine = mg.load_module("ref_refine_harmonic", os.path.join(base, "harmonic_pinn_simulation.py"))
    vanilla_fixture(refine, "m0_g10_64x3", [1, 64, 64, 64, 1], 400, 6, 0, 10.0, 3, 1.0)
    vanilla_fixture(refine, "m0_gm4_32x3", [1, 32, 32, 32, 1], 300, 7, 0, -4.0, 3, 1.0)
