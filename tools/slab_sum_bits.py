#!/usr/bin/env python3
"""Gradient and parameter BITS after k steps, for a few batch sizes / networks (slab counts 1 .. 512, with and without a remainder of the
eight-load groups of slab_column_sum): prints one sha256 per case.  Run it once per library build (GPE_HIP_LIB=...) and diff the outputs:
the unrolled slab-column sum must reproduce the plain loop's bits.   usage: tools/slab_sum_bits.py > bits.txt"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpe_pinn
from bench import reference_init

cases = [([1, 64, 64, 64, 64, 1], 65536), ([1, 64, 64, 64, 1], 4000), ([1, 64, 64, 64, 1], 37), ([2, 64, 64, 64, 64, 1], 300001),
         ([2, 128, 128, 128, 128, 128, 1], 50000), ([1, 32, 32, 32, 32, 1], 2048), ([2, 64, 64, 64, 64, 1], 1048576), ([3, 256, 256, 256, 1], 20000),
         ([2, 100, 100, 100, 1], 7800), ([1, 64, 64, 64, 1], 16384 * 3 + 5)]
for layers, N in cases:
  try:
    d = layers[0]
    rng = np.random.default_rng(N)
    x = (rng.random((N, d)) * 2 - 1).astype(np.float32) * 4
    xb = np.array([[-4.0] * d, [4.0] * d], np.float32)
    cfg = gpe_pinn.GPEConfig(layers=layers, gamma=10.0, p=3, kinetic_coeff=0.5, pot_scale=0.5, dx=8.0 ** d / N, w_bc=10.0, w_norm=20.0, lr=1e-3,
                             n_global=N, world_size=1)
    eng = gpe_pinn.Engine(cfg)
    eng.set_params(reference_init(layers, seed=1))
    eng.bind_points(torch.as_tensor(x, device="cuda")); eng.bind_boundary(torch.as_tensor(xb, device="cuda"))
    eng.step()
    g1 = np.asarray(eng.get_grad()).copy()
    eng.run(7)
    th = np.asarray(eng.get_params())
    k = eng.active_kernels
    print(layers, N, hashlib.sha256(g1.tobytes()).hexdigest()[:16], hashlib.sha256(th.tobytes()).hexdigest()[:16], k["bwd"][:40], flush=True)
    eng.close()
  except Exception as ex:          # noqa: BLE001 -- reported per case
    print(layers, N, "EXC", str(ex)[:200], flush=True)
