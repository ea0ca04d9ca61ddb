#!/usr/bin/env python3
"""Development check of the wide kernel set against the oracle (no forward_jets: works with -DGPE_FAST_BUILD libraries).
usage: wide_check.py [N] [H] [L]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpe_pinn
from oracle import gpe_oracle as go

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
Hh = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Lh = int(sys.argv[3]) if len(sys.argv) > 3 else 6
layers = [3] + [Hh] * Lh + [1]
kw = dict(layers=layers, gamma=1000.0, dx=0.004, omega=(1.0, 1.4, 2.0))
rng = np.random.default_rng(0)
x = rng.uniform(-3, 3, (N, 3)).astype(np.float32)
flat = (rng.normal(0, 1, go.param_count(layers)) * (0.1 if Hh > 128 else 0.15)).astype(np.float32)
x_bc = rng.uniform(-3, 3, (5, 3)).astype(np.float32)
pb = go.Problem(**kw)
t0 = time.time()
osc, ograd, ores = go.full_loss_and_grad(pb, flat.astype(np.float64), x.astype(np.float64), x_bc.astype(np.float64))
print("oracle %.1fs" % (time.time() - t0), flush=True)
res = {}
for name, path in (("generic", gpe_pinn.PATH_GENERIC), ("fused", gpe_pinn.PATH_FUSED)):
    cfg = gpe_pinn.GPEConfig(**kw, path=path)
    eng = gpe_pinn.Engine(cfg)
    eng.set_params(flat)
    eng.bind_points(torch.as_tensor(x, device="cuda"))
    eng.bind_boundary(torch.as_tensor(x_bc, device="cuda"))
    print(name, eng.active_kernels, flush=True)
    val = eng.forward(torch.as_tensor(x, device="cuda")).cpu().numpy()
    oj, _ = go.mlp_forward(go.unflatten(flat.astype(np.float64), layers), x.astype(np.float64), 0, value_only=True)
    print("  forward max err", np.abs(val - oj[0]).max(), "of", np.abs(oj[0]).max(), flush=True)
    rs, psi, r = eng.residual()
    print("  residual rel err", np.abs(r.cpu().numpy() - ores["residual"]).max() / np.abs(ores["residual"]).max(), flush=True)
    sc = eng.step()
    g = eng.get_grad()
    for k in ("mu", "loss", "pde", "bc", "norm"):
        print("  %-5s %.8g oracle %.8g rel %.2e" % (k, sc[k], osc[k], abs(sc[k] - osc[k]) / max(abs(osc[k]), 1e-30)))
    print("  grad rel err %.3e  |g| %.4g" % (np.abs(g - ograd).max() / np.abs(ograd).max(), np.linalg.norm(ograd)), flush=True)
    res[name] = g
    eng.close()
print("fused vs generic grad rel", np.abs(res["fused"] - res["generic"]).max() / np.abs(res["generic"]).max())
