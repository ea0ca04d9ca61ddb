#!/usr/bin/env python3
"""Per-phase cycle shares of the wide kernels (library built with -DGPE_STAMP; GPE_HIP_LIB points at it).
usage: GPE_HIP_LIB=build/variants/libgpe_stamp.so python tools/wide_stamps.py [workload] [points]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpe_pinn, bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg5_3d_6x256"]
x, dx, xb = bench.make_points(wl, 0, 1)
x = x[:int(sys.argv[2]) if len(sys.argv) > 2 else 131072]
cfg = gpe_pinn.GPEConfig(layers=wl["layers"], gamma=wl["gamma"], dx=dx, omega=tuple(wl.get("omega", (1.0, 1.0, 1.0))), lr=1e-3)
eng = gpe_pinn.Engine(cfg)
eng.set_params(bench.reference_init(wl["layers"]))
eng.bind_points(torch.as_tensor(x, device="cuda"))
eng.bind_boundary(torch.as_tensor(xb, device="cuda"))
eng.run(3); eng.synchronize()
out = (ctypes.c_ulonglong * 16)()
eng.lib.gpe_debug_read_wide_stamps(out)
eng.run(5); eng.synchronize()
rc = eng.lib.gpe_debug_read_wide_stamps(out)
v = np.array(list(out), dtype=np.float64)
names_f = ["epilogue+stores->top", "barrier1 wait", "AB write+barrier2", "MFMA loop", "-", "-", "-", "-"]
# w_bwd_map since round 3 (stamps 2, 3 unused): next tile's rows into ZB + loop top | barrier | adjoint products | activation adjoint,
# Zout, XT, prefetch | second barrier (single-buffer form only) | weight-gradient products
names_b = ["publish + top (st,w loads)", "barrier wait", "(r02: ZB write+transposes)", "(r02: barrier1 wait)", "adjoint MFMA", "act+Zout+XT+prefetch",
           "second barrier wait", "dW MFMA"]
print("rc", rc)
for nm, blk in (("w_forward", v[:8]), ("w_bwd_map", v[8:])):
    tot = blk.sum()
    print(nm, "total cycles (wave 0, all WGs) %.3e" % tot)
    for n, c in zip(names_f if nm == "w_forward" else names_b, blk):
        if c: print("   %-28s %6.2f %%" % (n, 100 * c / tot))

tr = (ctypes.c_ulonglong * 256)()
if eng.lib.gpe_debug_read_wide_trace(tr) == 0:
    t = np.array(list(tr), dtype=np.float64).reshape(2, 8, 16)
    for k, nm, ids in ((0, "w_forward (one layer of workgroup 0)", [0, 1, 2, 3]), (1, "w_bwd_map (one tile of workgroup 0)", [0, 1, 2, 3, 4, 5, 6, 7])):
        base = t[k][:, ids[0]].min()
        print(nm, ": stamp times per wave, cycles after the first wave's stamp 0")
        for w in range(8):
            print("   wave %d: " % w + " ".join("%7d" % (t[k][w][i] - base) for i in ids))
