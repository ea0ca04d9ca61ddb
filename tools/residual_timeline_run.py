#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gpe_pinn
os.environ.setdefault("GPE_GRAPH", "0")
N = 4000
x = np.linspace(0, 1, N).reshape(-1, 1).astype(np.float32)
cfg = gpe_pinn.GPEConfig(layers=[1, 64, 64, 64, 1], net_kind=gpe_pinn.capi.NET_RESIDUAL, activation=1, kinetic_coeff=1.0, potential=gpe_pinn.POT_GAUSSIAN,
                         pot_a=0.5, gamma=5.0, p=4, base_mode=0, base_kind=gpe_pinn.capi.BASE_BOX, envelope=gpe_pinn.capi.ENV_SIN, perturb_scale=0.01,
                         dx=1.0 / (N - 1), lr=1e-3, sched=gpe_pinn.SCHED_COSINE_LOSS)
eng = gpe_pinn.Engine(cfg)
torch.manual_seed(0)
eng.set_params((torch.randn(eng.n_params) * 0.1).numpy())
eng.bind_points(torch.as_tensor(x, device="cuda"))
eng.bind_boundary(torch.tensor([[0.0], [1.0]], device="cuda"))
eng.run(60); eng.synchronize()
