#!/usr/bin/env python3
"""train_pinn at the reference script's own size (src/gross_pitaevskii_2D_minimal.py:373-381: N_u = 500, N_f = 10 000 draws, [2,100,100,100,1],
2001 epochs) and at the function's default network ([2,400,400,400,1], :278): wall time per epoch on the engine.
usage: python tools/pinn2d_reference_size.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpe_pinn

gpe_pinn.pinn2d_minimal.train_pinn(N_u=50, N_f=500, layers=[2, 64, 64, 1], epochs=50, verbose=False).close()       # (device warm-up)
for layers, epochs in (([2, 100, 100, 100, 1], 2001), ([2, 400, 400, 400, 1], 1000), ([2, 128, 128, 128, 1], 2001), ([2, 64, 64, 64, 1], 2001)):
    torch.manual_seed(0)
    np.random.seed(0)
    t0 = time.perf_counter()
    model = gpe_pinn.pinn2d_minimal.train_pinn(N_u=500, N_f=10000, layers=layers, epochs=epochs, verbose=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = model.history
    eng = model._get_engine()
    print(f"{str(layers):24s} {epochs} epochs in {dt:6.2f} s = {dt / epochs * 1e6:8.1f} us/epoch   loss {h['loss'][0]:.4g} -> {h['loss'][-1]:.4g}   "
          f"lambda {h['mu'][0]:.4g} -> {h['mu'][-1]:.4g}   path {eng.active_path} {eng.active_kernels if hasattr(eng, 'n_local') else ''}", flush=True)
    model.close()
