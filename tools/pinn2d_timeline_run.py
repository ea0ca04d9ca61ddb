#!/usr/bin/env python3
"""200 epochs of train_pinn at the reference script's own size ([2,100,100,100,1], 10 000 draws) -- for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gpe_pinn
torch.manual_seed(0); np.random.seed(0)
layers = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 100, 100, 100, 1]
os.environ.setdefault("GPE_GRAPH", "0")
m = gpe_pinn.pinn2d_minimal.train_pinn(N_u=500, N_f=10000, layers=layers, epochs=200, verbose=False)
m.close()
