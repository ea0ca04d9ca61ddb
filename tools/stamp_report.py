#!/usr/bin/env python3
"""Run the NS workload on the -DGPE_STAMP build and print the per-phase cycle shares of the reverse kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPE_HIP_LIB", os.path.join(ROOT, "build/variants/libgpe_stamp.so"))
import numpy as np, torch
import bench, gpe_pinn
wl = bench.WORKLOADS["ns_2d_4x64"]
x, dx, xb = bench.make_points(wl, 0, 1)
eng = gpe_pinn.Engine(gpe_pinn.GPEConfig(layers=wl["layers"], gamma=wl["gamma"], dx=dx, w_bc=0.0))
eng.set_params(bench.reference_init(wl["layers"]))
eng.bind_points(torch.as_tensor(x, device="cuda"))
eng.run(2)
out = (ctypes.c_ulonglong * 16)()
eng.lib.gpe_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
eng.lib.gpe_debug_read_stamps(eng._h, out)
eng.run(3)
eng.lib.gpe_debug_read_stamps(eng._h, out)
v = np.array(list(out)[:8], dtype=np.float64)
names = ["0 seeds+output layer", "1 bias grads", "2 Zb transposes", "3 X load+recompute+transpose", "4 B2 MFMA+LDS add",
         "5 B1 MFMA+adjoint+copy", "6 layer-0 grads", "7 X load+recompute+adjoint (3 = its transposes)"]
tot = v.sum()
for n, c in zip(names, v):
    print("%-32s %6.2f %%   %.3e wave-cycles" % (n, 100 * c / tot, c))
print("total wave-cycles %.3e ; per tile %.0f" % (tot, tot / (3 * 65536)))
