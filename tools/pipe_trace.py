#!/usr/bin/env python3
"""Phase timeline of f_backward_pipe (library built with -DGPE_STAMP: tools/build_variant.sh pstamp -DGPE_STAMP).
Prints the per-wave phase shares (barrier wait / products / VALU+LDS part) over all workgroups and, for a few CUs, the stamps of
two tile iterations of both resident workgroups side by side.   usage: GPE_HIP_LIB=build/variants/libgpe_pstamp.so python tools/pipe_trace.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPE_HIP_LIB", os.path.join(ROOT, "build/variants/libgpe_pstamp.so"))
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
NST = 3
eng.run(NST)
eng.lib.gpe_debug_read_stamps(eng._h, out)
v = np.array(list(out), dtype=np.float64).reshape(4, 4)[:, :3]
print("phase sums per wave slot (cycles per tile): barrier wait | products | VALU+LDS part")
for w in range(4):
    per = v[w] / (NST * 65536)
    print("  wave %d: %8.0f %8.0f %8.0f   total %8.0f  (products %.3f)" % (w, per[0], per[1], per[2], per.sum(), per[1] / per.sum()))
n = 512 * 4 * 32
tr = (ctypes.c_ulonglong * n)()
eng.lib.gpe_debug_read_trace.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
rc = eng.lib.gpe_debug_read_trace(eng._h, tr, n)
t = np.array(list(tr), dtype=np.uint64).reshape(512, 4, 32)
hw = t[:, :, 0]
cu = {}
for b in range(512):
    h = int(hw[b, 0]) & 0xffffffff; xcc = int(hw[b, 0]) >> 32
    key = (xcc & 0xf, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xf)
    cu.setdefault(key, []).append(b)
print("distinct (xcc, se, sh, cu):", len(cu), " workgroups per CU:", sorted(set(len(v) for v in cu.values())))
lo = t[:, 0, 25].astype(np.int64); hi = t[:, 0, 26].astype(np.int64)
print("tile-loop lifetime of a workgroup, cycles: min %.3e  mean %.3e  max %.3e" % ((hi - lo).min(), (hi - lo).mean(), (hi - lo).max()))
both = []; span = []
for key, blocks in cu.items():
    if len(blocks) == 2:
        a, b = blocks
        both.append(max(0, min(hi[a], hi[b]) - max(lo[a], lo[b]))); span.append(max(hi[a], hi[b]) - min(lo[a], lo[b]))
both = np.array(both, dtype=np.float64); span = np.array(span, dtype=np.float64)
print("per CU: span of its two workgroups mean %.3e max %.3e ; both resident %.3f of the span (min %.3f)" %
      (span.mean(), span.max(), (both / span).mean(), (both / span).min()))
shown = 0
for key, blocks in sorted(cu.items()):
    if len(blocks) != 2 or shown >= 3:
        continue
    shown += 1
    st = t[blocks][:, :, 1:25].astype(np.int64)          # [2 wg][4 waves][24]
    base = st[st > 0].min()
    print("CU", key, "workgroups", blocks, " simd of waves:", [[(int(hw[b, w]) >> 4) & 3 for w in range(4)] for b in blocks])
    for it in range(2):
        for k in range(3):
            for e, nm in enumerate(("at barrier", "released  ", "products done")):
                row = []
                for bi in range(2):
                    row.append(" ".join("%7d" % (st[bi, w, 12 * it + 4 * k + e] - base) for w in range(4)))
                print("   it %d interval %d %-13s | %s | %s" % (it, k, nm, row[0], row[1]))
