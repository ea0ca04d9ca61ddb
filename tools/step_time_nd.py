#!/usr/bin/env python3
"""Wall time per step of gpe_run for any network / batch size (harmonic trap, north-star loss):  step_time_nd.py 2,128,128,128,128,128,1 16384 [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpe_pinn
layers = [int(v) for v in sys.argv[1].split(",")]
N = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
d = layers[0]
rng = np.random.default_rng(0)
x = rng.uniform(-6, 6, (N, d)).astype(np.float32)
cfg = gpe_pinn.GPEConfig(layers=layers, gamma=100.0, dx=float(12.0 ** d / N), lr=1e-3, complex_psi=layers[-1] == 2)
eng = gpe_pinn.Engine(cfg)
torch.manual_seed(0)
eng.set_params((torch.randn(eng.n_params) * 0.1).numpy())
eng.bind_points(torch.as_tensor(x, device="cuda"))
eng.run(30); eng.synchronize()
t = []
for _ in range(3):
    t0 = time.perf_counter(); eng.run(steps); eng.synchronize(); t.append((time.perf_counter() - t0) / steps * 1e6)
k = eng.active_kernels
print("%s N=%d: %.1f us/step  %.3g points/s  fwd=%s bwd=%s" % (sys.argv[1], N, min(t), N / min(t) * 1e6, k["fwd"][:28], k["bwd"][:28]))
