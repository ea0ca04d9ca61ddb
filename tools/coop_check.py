#!/usr/bin/env python3
"""Quick parity check of the reverse-kernel variant selected by the environment (GPE_COOP, GPE_STAGE_MIN_TILES, GPE_HIP_LIB)
against the fp64 oracle: 2D [2,64,64,64,64,1] (works with -DGPE_FAST_BUILD libraries)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gpe_pinn
from oracle import gpe_oracle as go
from tests.test_gpu_parity import make_engine, _inputs
from tests import helpers as H
for layers, N in (([2, 64, 64, 64, 64, 1], 777), ([2, 64, 64, 64, 1], 4097), ([2, 64, 64, 1], 5)):
    kw = dict(layers=layers, gamma=50.0, dx=0.01)
    x, flat, x_bc = _inputs(kw, N)
    pb = go.Problem(**kw)
    osc, ograd, ores = go.full_loss_and_grad(pb, flat.astype(np.float64), x.astype(np.float64), x_bc.astype(np.float64))
    eng = make_engine(pb, flat, x, x_bc)
    sc = eng.step()
    g = eng.get_grad()
    print(layers, N, "loss %.6f/%.6f mu %.6f/%.6f GRAD rel err %.2e  |g| %.4f/%.4f" % (sc["loss"], osc["loss"], sc["mu"], osc["mu"],
          H.rel_err(g, ograd), np.linalg.norm(g), np.linalg.norm(ograd)))
    # per-block errors
    off = 0
    for li in range(len(layers) - 1):
        nW = layers[li] * layers[li + 1]; nb = layers[li + 1]
        print("   W%d %.1e  b%d %.1e" % (li, H.rel_err(g[off:off + nW], ograd[off:off + nW]), li, H.rel_err(g[off + nW:off + nW + nb], ograd[off + nW:off + nW + nb])))
        off += nW + nb
    eng.close()
