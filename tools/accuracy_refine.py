#!/usr/bin/env python3
"""BASELINE metric, second half: ground-state eigenvalue error.  Runs the reference's headline experiment
(refine/harmonic_pinn_simulation.py:963-1009: PL-PINN, mode 0, p = 3, gamma = 0, 0.5, ..., gamma_max, 5001 epochs per
stage with early stopping at tol = 1e-5, lr = 1e-3, N_f points on [-10,10], [1,64,64,64,1]) on the engine and compares
lambda(gamma) with the independent fp64 solver oracle/gp_ground_state.py (checker only)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gpe_pinn
from gpe_pinn import refine
from oracle import gp_ground_state as gs

ap = argparse.ArgumentParser()
ap.add_argument("--gamma-max", type=float, default=100.0)
ap.add_argument("--alpha", type=float, default=0.5)
ap.add_argument("--n", type=int, default=4000)
ap.add_argument("--epochs", type=int, default=5001)
ap.add_argument("--tol", type=float, default=1e-5)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "accuracy_refine.json"))
a = ap.parse_args()
torch.manual_seed(0)
lb, ub = -10, 10
X = np.linspace(lb, ub, a.n).reshape(-1, 1)
gammas = [k * a.alpha for k in range(int(round(a.gamma_max / a.alpha)) + 1)]
t0 = time.time()
models, mu_table, hist, const, epochs = refine.train_gpe_model(gammas, [0], 3, X, lb, ub, [1, 64, 64, 64, 1], a.epochs, a.tol,
                                                             0.01, potential_type="harmonic", lr=1e-3, verbose=False)
wall = time.time() - t0
check = [g for g in (0.0, 10.0, 20.0, 40.0, 60.0, 80.0, 100.0) if g <= a.gamma_max]
exact, _ = gs.ground_state_1d(check, c=1.0, vscale=1.0)
mu = dict(mu_table[0])
rows = []
for g in check:
    m = models[0][g]
    rows.append(dict(gamma=g, lam_engine_recorded=mu[g], lam_engine_last=m.last_mu, lam_exact=float(exact[g]),
                     abs_err=abs(m.last_mu - float(exact[g])), epochs=epochs[0][g]))
    print(rows[-1], flush=True)
total_epochs = int(sum(min(v + 1, a.epochs) for v in epochs[0].values()))
out = dict(experiment="refine PL-PINN gamma continuation, mode 0, p=3", n_points=a.n, stages=len(gammas), epochs_per_stage=a.epochs,
           total_epochs=total_epochs, wall_seconds=wall, tol=a.tol, rows=rows, max_abs_err=max(r["abs_err"] for r in rows))
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(out, open(a.out, "w"), indent=1)
print("wall %.1f s, %d epochs, max |lambda - exact| = %.2e" % (wall, total_epochs, out["max_abs_err"]))
