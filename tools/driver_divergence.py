#!/usr/bin/env python3
"""Calibration of the driver tests' trajectory bounds: the refine / notebook drivers on the engine, every stage's
per-epoch history against the fp64 oracle stepped from the stage's own start weights.  Prints max relative loss error and max |mu error|
over epochs [0, k) for k in 1, 2, 5, 10, 20, 40, 80."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpe_pinn
from gpe_pinn import refine, notebook
from oracle import gpe_oracle as go
from tests import helpers as H

KS = (1, 2, 5, 10, 20, 40, 80)


def show(tag, dl, dm):
    print("%-34s " % tag + "  ".join("k<%d: %.1e / %.1e" % (k, dl[:k].max(), dm[:k].max()) for k in KS if k <= len(dl)), flush=True)


for name in ("fx_refdriver_m0_3stages.npz", "fx_refdriver_m1_2stages.npz", "fx_refdriver_m0_earlystop.npz"):
    fx = H.load_fx(name)
    layers = [int(v) for v in fx["layers"]]
    N, epochs, tol = int(fx["N"]), int(fx["epochs"]), float(fx["tol"])
    gammas = [float(g) for g in fx["gammas"]]
    mode = int(fx["modes"][0])
    torch.manual_seed(int(fx["seed"]))
    lb, ub = -10, 10
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    models, mu_table, hist, const, ep = refine.train_gpe_model(gammas, [mode], 3, X, lb, ub, layers, epochs, tol, 0.01,
                                                               potential_type="harmonic", lr=1e-3, verbose=False)
    for g in gammas:
        m = models[mode][g]
        pb = go.Problem(layers=layers, activation=1, kinetic_coeff=1.0, potential=go.POT_HARMONIC, pot_scale=1.0, gamma=g, p=3,
                        base_mode=mode, base_deriv=0, perturb_scale=0.01 / float(const[mode]), bc_nn_scale=1.0, w_bc=10.0, w_norm=20.0,
                        dx=float(X[1, 0] - X[0, 0]))
        dl, dm, dlr = H.stage_divergence(pb, m.start_flat, X, np.array([[lb], [ub]], float), m.history, 80, go.SCHED_COSINE_LOSS, 1e-3,
                                         T_0=200.0, T_mult=2.0, eta_min=1e-6)
        show(f"{name[12:-4]} gamma={g} (mu {mu_table[mode][gammas.index(g)][1]:.5f}, stop {ep[mode][g]})", dl, dm)

fx = H.load_fx("fx_nbdriver_small.npz")
layers = [int(v) for v in fx["layers"]]
N, epochs = int(fx["N"]), int(fx["epochs"])
torch.manual_seed(int(fx["seed"]))
lb, ub = -10, 10
X = np.linspace(lb, ub, N).reshape(-1, 1)
models, mu_table = notebook.train_gpe_model([1], [2, 3], [0, 1], X, lb, ub, layers, epochs, potential_type="harmonic", lr=1e-3, verbose=False)
for mode in (0, 1):
    for power in (2, 3):
        m = models[mode][power]
        pb = go.Problem(layers=layers, activation=0, kinetic_coeff=0.5, potential=go.POT_HARMONIC, pot_scale=0.5, gamma=1.0, p=power,
                        base_mode=mode, base_deriv=1, perturb_scale=1.0, bc_nn_scale=1.0, w_bc=10.0, w_norm=20.0, w_sym=5.0,
                        sym_sign=(-1.0 if mode % 2 == 1 else 1.0), dx=float(X[1, 0] - X[0, 0]))
        dl, dm, dlr = H.stage_divergence(pb, m.start_flat, X, np.array([[lb], [ub]], float), m.history, 80, go.SCHED_PLATEAU, 1e-3,
                                         factor=0.5, patience=100, min_lr=1e-5)
        show(f"notebook mode {mode} p={power} (mu {dict(mu_table[mode])[power]:.5f})", dl, dm)
