import cProfile, pstats, sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gpe_pinn
from gpe_pinn import refine
X = np.linspace(-10, 10, 4000).reshape(-1, 1)
gammas = [0.5 * k for k in range(21)]
torch.manual_seed(0)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
out = refine.train_gpe_model(gammas, [0], 3, X, -10, 10, [1, 64, 64, 64, 1], 5001, 1e-7, 0.01, potential_type="harmonic", lr=1e-3, verbose=False)
pr.disable()
print("wall", time.time() - t0, "epochs", sum(len(v) for v in out[2].values()) if isinstance(out[2], dict) else "?")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
