#!/usr/bin/env python3
"""Step time at the reference's own problem size (N_f = 4000, [1,64,64,64,1], refine flavour): wall per step of gpe_run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpe_pinn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
x = np.linspace(-10, 10, N).reshape(-1, 1).astype(np.float32)
cfg = gpe_pinn.GPEConfig(layers=[1, 64, 64, 64, 1], activation=1, kinetic_coeff=1.0, pot_scale=1.0, gamma=10.0, base_mode=0,
                         perturb_scale=0.02, dx=20.0 / (N - 1), lr=1e-3, sched=gpe_pinn.SCHED_COSINE_LOSS)
eng = gpe_pinn.Engine(cfg)
torch.manual_seed(0)
eng.set_params((torch.randn(eng.n_params) * 0.2).numpy())
eng.bind_points(torch.as_tensor(x, device="cuda"))
eng.bind_boundary(torch.tensor([[-10.0], [10.0]], device="cuda"))
eng.run(50); eng.synchronize()
t0 = time.perf_counter(); eng.run(steps); t_enq = time.perf_counter() - t0; eng.synchronize(); dt = time.perf_counter() - t0
print("N=%d: %.1f us/step (%d steps), %.3g points/s; host enqueue %.1f us/step" % (N, dt / steps * 1e6, steps, N * steps / dt, t_enq / steps * 1e6))
