import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, gpe_pinn
from gpe_pinn import GPEConfig, Engine
x = torch.linspace(-10, 10, 4000, device="cuda").reshape(-1, 1).contiguous()
xb = torch.tensor([[-10.0], [10.0]], device="cuda")
cfg = GPEConfig(layers=[1, 64, 64, 64, 1], activation=1, kinetic_coeff=1.0, pot_scale=1.0, base_mode=0, history_capacity=5001)
e = Engine(cfg); e.close()
torch.cuda.synchronize()
t = {k: 0.0 for k in ("create", "setp", "bind", "run100", "hist", "getp", "close")}
n = 30
flat = np.zeros(8513, np.float32) + 0.01
for _ in range(n):
    t0 = time.perf_counter(); e = Engine(cfg); t1 = time.perf_counter(); t["create"] += t1 - t0
    e.set_params(flat); t2 = time.perf_counter(); t["setp"] += t2 - t1
    e.bind_points(x); e.bind_boundary(xb); t3 = time.perf_counter(); t["bind"] += t3 - t2
    e.run(100); e.synchronize(); t4 = time.perf_counter(); t["run100"] += t4 - t3
    h = e.read_history(1, 100); t5 = time.perf_counter(); t["hist"] += t5 - t4
    p = e.get_params(); t6 = time.perf_counter(); t["getp"] += t6 - t5
    e.close(); t7 = time.perf_counter(); t["close"] += t7 - t6
print({k: round(v / n * 1e3, 3) for k, v in t.items()}, "ms each")
