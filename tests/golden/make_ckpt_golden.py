#!/usr/bin/env python3
"""Golden vectors from the reference's STORED artefact
  Gross-Pitaevskii/src/final/refine/harmonic_test/harmonic_mode_zero_plot_data.pkl
(trained PL-PINN weights for modes 0..5 at gamma = 0 and the mu_table the reference recorded for them; producer
refine/plot_harmonic_potential_at_ground_state.py:1258-1309: N_f = 4000 on [-10,10], q = 0.01, [1,64,64,64,1]).
Read with the package's restricted unpickler (no arbitrary globals).  Runs only in the build container.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib
ck = importlib.import_module("gross-pitaevskii-eigenvalue-problem_amd.checkpoint")

REFDIR = "/root/reference/Gross-Pitaevskii/src/final/refine"
d = ck.load_results("harmonic_mode_zero_plot_data.pkl", os.path.join(REFDIR, "harmonic_test"))
out = dict(layers=np.array(d["models_state_dicts"][0][0]["layers"]), N=4000, lb=-10.0, ub=10.0, perturb_const=0.01)
modes = sorted(d["models_state_dicts"])
out["modes"] = np.array(modes)
for mode in modes:
    md = d["models_state_dicts"][mode][0]
    sd = md["state_dict"]
    out[f"flat_mode{mode}"] = np.concatenate([sd[k].numpy().ravel() for k in sd]).astype(np.float32)
    out[f"mu_mode{mode}"] = float(d["mu_table"][mode][0][1])
    out[f"const_mode{mode}"] = float(d["constant_history"][mode])
    out[f"epochs_mode{mode}"] = int(d["epochs_history"][mode][0])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fx_ckpt_harmonic_modes.npz"), **out)
print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})

# ---- box_test/box_mode_zero_plot_data.pkl (refine/plot_box_potential_at_ground_state.py): modes 0,1 at gamma = 0 -------------
d = ck.load_results("box_mode_zero_plot_data.pkl", os.path.join(REFDIR, "box_test"))
out = dict(layers=np.array(d["models_state_dicts"][0][0]["layers"]), N=4000, lb=0.0, ub=1.0, perturb_const=0.01)
modes = sorted(d["models_state_dicts"])
out["modes"] = np.array(modes)
for mode in modes:
    md = d["models_state_dicts"][mode][0]
    sd = md["state_dict"]
    out[f"flat_mode{mode}"] = np.concatenate([sd[k].numpy().ravel() for k in sd]).astype(np.float32)
    out[f"mu_mode{mode}"] = float(d["mu_table"][mode][0][1])
    out[f"const_mode{mode}"] = float(d["constant_history"][mode])
    out[f"epochs_mode{mode}"] = int(d["epochs_history"][mode][0])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fx_ckpt_box_modes.npz"), **out)
print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})
