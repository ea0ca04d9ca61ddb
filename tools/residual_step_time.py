#!/usr/bin/env python3
"""Step time of the residual-block network of refine/box_to_gaussian_pinn_simulation.py at its own size (N_f = 4000, layers [1,64,64,64,1] =
Linear + 2 residual blocks + Linear, ShiftedTanh first layer, p = 16, Gaussian potential) beside the plain MLP with the same number of maps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpe_pinn
N = 4000
x = np.linspace(0, 1, N).reshape(-1, 1).astype(np.float32)
for name, kw in (("residual [1,64,64,64,1] (2 blocks)", dict(layers=[1, 64, 64, 64, 1], net_kind=gpe_pinn.capi.NET_RESIDUAL)),
                 ("plain MLP [1,64,64,64,64,64,1]", dict(layers=[1, 64, 64, 64, 64, 64, 1])),
                 ("plain MLP [1,64,64,64,1]", dict(layers=[1, 64, 64, 64, 1]))):
    cfg = gpe_pinn.GPEConfig(activation=1, kinetic_coeff=1.0, potential=gpe_pinn.POT_GAUSSIAN, pot_a=0.5, gamma=5.0, p=4, base_mode=0,
                             base_kind=gpe_pinn.capi.BASE_BOX, envelope=gpe_pinn.capi.ENV_SIN, perturb_scale=0.01, dx=1.0 / (N - 1), lr=1e-3,
                             sched=gpe_pinn.SCHED_COSINE_LOSS, **kw)
    eng = gpe_pinn.Engine(cfg)
    torch.manual_seed(0)
    eng.set_params((torch.randn(eng.n_params) * 0.1).numpy())
    eng.bind_points(torch.as_tensor(x, device="cuda"))
    eng.bind_boundary(torch.tensor([[0.0], [1.0]], device="cuda"))
    eng.run(50); eng.synchronize()
    t0 = time.perf_counter(); eng.run(1000); eng.synchronize(); dt = (time.perf_counter() - t0) / 1000 * 1e6
    print(f"{name:38s} {dt:8.1f} us/step   path {eng.active_path}  {eng.active_kernels['fwd'][:30]} / {eng.active_kernels['bwd'][:30]}", flush=True)
    eng.close()
