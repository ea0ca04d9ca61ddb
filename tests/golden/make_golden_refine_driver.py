#!/usr/bin/env python3
"""
tests/golden/make_golden_refine_driver.py -- pin the CONTINUATION DRIVER to the reference's own code: imports
/root/reference/Gross-Pitaevskii/src/final/refine/harmonic_pinn_simulation.py (read-only) and runs ITS train_gpe_model
(:220-430, with its pre-training :650-701, cosine(loss) scheduler, early stopping :389-400) for a short seeded schedule.
Stores what the function returns: mu_table, epochs_history, constant_history, the history arrays, final weights per gamma.
Runs only in the build container; the .npz is committed (fx_refdriver_*.npz).

With --study it also runs a longer schedule (gamma = 0, 2, ..., 10, up to 5001 epochs per stage, the reference's literal
tol = 1e-5, N = 1000) and writes profiles/r02/reference_refine_tol1e-5_cpu.json: the reference's OWN |lambda - lambda_exact| at
that stopping tolerance, next to the fp64 finite-difference solver (answers whether errors of O(1e-3) at tol = 1e-5 are the
reference's behaviour or the engine's).

Usage:  MPLBACKEND=Agg python tests/golden/make_golden_refine_driver.py [--study]
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
torch.set_num_threads(4)


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def flat_params(model):
    return np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]).astype(np.float32)


def run(refine, gammas, modes, N, layers, epochs, tol, seed, perturb_const=0.01, lr=1e-3):
    torch.manual_seed(seed)
    lb, ub = -10, 10
    X = np.linspace(lb, ub, N).reshape(-1, 1)
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        out = refine.train_gpe_model(gammas, modes, 3, X, lb, ub, layers, epochs, tol, perturb_const,
                                     potential_type='harmonic', lr=lr, verbose=False)
    return out, time.time() - t0


def fixture(refine, tag, gammas, modes, N, layers, epochs, tol, seed):
    (models, mu_table, hist, const, ep), wall = run(refine, gammas, modes, N, layers, epochs, tol, seed)
    fx = dict(layers=np.array(layers), N=N, seed=seed, epochs=epochs, tol=tol, gammas=np.array(gammas, dtype=np.float64),
              modes=np.array(modes), perturb_const=0.01, lr=1e-3, wall_seconds=wall)
    for mode in modes:
        fx[f"mu_mode{mode}"] = np.array(mu_table[mode], dtype=np.float64)                 # rows (gamma, final_mu)
        fx[f"const_mode{mode}"] = float(const[mode])
        fx[f"epochs_mode{mode}"] = np.array([ep[mode][g] for g in gammas], dtype=np.int64)
        for g in gammas:
            h = hist[mode][g]
            fx[f"loss_mode{mode}_g{g}"] = np.array(h['loss'], dtype=np.float64)
            fx[f"lambda_mode{mode}_g{g}"] = np.array(h['lambda'], dtype=np.float64)
            fx[f"constraint_mode{mode}_g{g}"] = np.array(h['constraint'], dtype=np.float64)
            fx[f"flat_mode{mode}_g{g}"] = flat_params(models[mode][g])
    np.savez_compressed(os.path.join(OUT, f"fx_refdriver_{tag}.npz"), **fx)
    print("wrote refdriver", tag, {m: mu_table[m] for m in modes}, {m: ep[m] for m in modes}, f"{wall:.0f} s", flush=True)


if __name__ == "__main__":
    refine = load_module("ref_refine_harmonic", os.path.join(REF, "Gross-Pitaevskii/src/final/refine/harmonic_pinn_simulation.py"))
    if "--study" not in sys.argv:
        fixture(refine, "m0_3stages", [0.0, 0.5, 1.0], [0], 400, [1, 64, 64, 64, 1], 600, 1e-5, 0)
        fixture(refine, "m1_2stages", [0.0, 1.0], [1], 300, [1, 32, 32, 32, 1], 400, 1e-5, 1)
        fixture(refine, "m0_earlystop", [0.0, 0.5, 1.0], [0], 400, [1, 64, 64, 64, 1], 600, 1.5e-3, 0)      # loose tol: stages stop early
    else:
        sys.path.insert(0, ROOT)
        from oracle import gp_ground_state as gs
        gammas = [0.0, 2.0, 4.0, 6.0, 8.0, 10.0]
        (models, mu_table, hist, const, ep), wall = run(refine, gammas, [0], 1000, [1, 64, 64, 64, 1], 5001, 1e-5, 0)
        exact, _ = gs.ground_state_1d(gammas, c=1.0, vscale=1.0)
        rows = [dict(gamma=g, lam_reference=float(mu), lam_exact=float(exact[g]), abs_err=abs(float(mu) - float(exact[g])),
                     epochs=int(ep[0][g])) for g, mu in mu_table[0]]
        out = dict(what="the REFERENCE's own train_gpe_model (imported, CPU, seed 0): gamma continuation in steps of 2, N = 1000, "
                        "tol = 1e-5 (its literal), <= 5001 epochs per stage; lambda_exact from oracle/gp_ground_state.py",
                   wall_seconds=wall, rows=rows, max_abs_err=max(r["abs_err"] for r in rows))
        os.makedirs(os.path.join(ROOT, "profiles", "r02"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "profiles", "r02", "reference_refine_tol1e-5_cpu.json"), "w"), indent=1)
        for r in rows:
            print(r)
