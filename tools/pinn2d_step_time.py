#!/usr/bin/env python3
"""Step time of the reference's 2D class loss (src/gross_pitaevskii_2D.py:215-242: energy-functional lambda, two regularisers, Riesz sum,
10 x boundary mean) on the engine at the north-star size, beside the north-star loss on the SAME network and points (Rayleigh lambda,
normalisation term): what the non-fusable terms cost (k_head_pde / k_seed_pde as their own launches instead of riding in the forward /
reverse kernels).   usage: python tools/pinn2d_step_time.py [N_f] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpe_pinn
from gpe_pinn import GPEConfig, Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
layers = [2, 64, 64, 64, 64, 1]
torch.manual_seed(0)
np.random.seed(0)
model = gpe_pinn.pinn2d.GrossPitaevskiiPINN(layers, g=500.0)
model.apply(gpe_pinn.pinn2d.initialize_weights)
X_f, X_u, _ = gpe_pinn.pinn2d.prepare_training_data(500, N)
xf = torch.as_tensor(X_f.astype(np.float32), device="cuda")
xu = torch.as_tensor(X_u.astype(np.float32), device="cuda")


def timed(eng, label):
    eng.run(5)
    eng.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.run(steps)
        eng.synchronize()
        t.append((time.perf_counter() - t0) / steps * 1e3)
    eng.profile_enable(True)
    eng.run(steps)
    eng.synchronize()
    pr = eng.profile_read()
    eng.profile_enable(False)
    sc = eng.read_scalars()
    ms = float(np.median(t))
    print(f"{label:34s} {ms:8.4f} ms/step  {N / ms * 1e3:10.4e} points/s  fwd {pr['fwd_ms'] / max(pr['fwd_launches'], 1):.4f} ms  "
          f"bwd {pr['bwd_ms'] / max(pr['bwd_launches'], 1):.4f} ms  kernels {eng.active_kernels}  loss {sc['loss']:.5g} mu {sc['mu']:.5g}", flush=True)
    return ms


eng = model._get_engine()
eng.set_loss_weights(*model._W_FULL)
model._bind(eng, xf, xu)
a = timed(eng, "2D class loss (pinn2d)")
flat = model._flat.copy()
model.close()
cfg = GPEConfig(layers=layers, gamma=500.0, kinetic_coeff=1.0, potential=gpe_pinn.POT_PRECOMPUTED, p=3, w_bc=10.0, w_norm=20.0,
                dx=float(np.pi * (np.pi / 2) ** 2 / N), lr=1e-3, clip_norm=0.0)
eng = Engine(cfg)
eng.set_params(flat)
eng.bind_points(xf, model.compute_potential(xf))
eng.bind_boundary(xu)
b = timed(eng, "same points, north-star loss")
print(f"class loss / north-star loss step time: {a / b:.4f}")
