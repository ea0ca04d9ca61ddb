#!/usr/bin/env python3
"""f_forward_b6 / f_backward_coop<..., B6> (six bf16 MFMA products per fp32 product) against the fp32-MFMA kernels and the fp64 oracle: one training step.
usage: [GPE_HIP_LIB=...] python tools/b6_check.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import gpe_pinn
from oracle import gpe_oracle as go
import helpers as H
from test_gpu_parity import make_engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3001
kw = dict(layers=[2, 64, 64, 64, 64, 1], gamma=500.0, dx=36 / N)
rng = np.random.default_rng(0)
x = rng.uniform(-3, 3, (N, 2)).astype(np.float32)
flat = (rng.normal(0, 1, go.param_count(kw["layers"])) * 0.3).astype(np.float32)
xb = rng.uniform(-3, 3, (5, 2)).astype(np.float32)
pb = go.Problem(**kw)
osc, ograd, _ = go.full_loss_and_grad(pb, flat.astype(np.float64), x.astype(np.float64), xb.astype(np.float64))
os.environ["GPE_COOP_FWD_MAX_TILES"] = "0"
res = {}
for b6, r6 in (("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")):
    os.environ["GPE_FWD_B6"] = b6
    os.environ["GPE_BWD_B6"] = r6
    eng = make_engine(pb, flat, x, xb, path=gpe_pinn.PATH_FUSED)
    print("fwd b6 =", b6, " bwd b6 =", r6, eng.active_kernels)
    sc = eng.step()
    g = eng.get_grad()
    res[b6 + r6] = (sc, g)
    print("   loss %.8e (oracle %.8e, rel %.2e)  mu %.7f (oracle %.7f)  grad rel err vs oracle %.2e" %
          (sc["loss"], osc["loss"], abs(sc["loss"] - osc["loss"]) / abs(osc["loss"]), sc["mu"], osc["mu"], H.rel_err(g, ograd)))
    eng.close()
for k in ("10", "01", "11"):
    print("fwd/bwd b6 = %s vs fp32 kernels: grad rel diff %.2e, loss rel diff %.2e" %
          (k, H.rel_err(res[k][1], res["00"][1]), abs(res[k][0]["loss"] - res["00"][0]["loss"]) / abs(res["00"][0]["loss"])))
