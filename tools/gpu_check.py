#!/usr/bin/env python3
"""First-contact diagnostic on a GPU box: error table of engine (both kernel sets) vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpe_pinn
from gpe_pinn import GPEConfig, Engine
from oracle import gpe_oracle as go


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def run_case(name, kw, N, path, seed=0, xbc=True, scale=0.3):
    rng = np.random.default_rng(seed)
    layers = kw["layers"]; d = layers[0]
    if d == 1:
        x = np.linspace(-6, 6, N).reshape(-1, 1)
    else:
        x = rng.uniform(-3, 3, (N, d))
    x = x.astype(np.float32)
    P = go.param_count(layers)
    flat = (rng.normal(0, 1, P) * scale).astype(np.float32)
    x_bc = None
    if xbc:
        x_bc = (np.array([[-6.0], [6.0]]) if d == 1 else rng.uniform(-3, 3, (5, d))).astype(np.float32)
    pb = go.Problem(**kw)
    t0 = time.time()
    osc, ograd, ores = go.full_loss_and_grad(pb, flat.astype(np.float64), x.astype(np.float64),
                                             None if x_bc is None else x_bc.astype(np.float64))
    params = go.unflatten(flat.astype(np.float64), layers)
    ojets, _ = go.mlp_forward(params, x.astype(np.float64), pb.activation)
    cfg = GPEConfig(**kw, path=path, lr=1e-3, w_bc=pb.w_bc if xbc else 0.0) if False else GPEConfig(**{**kw, "path": path})
    if not xbc:
        cfg.w_bc = 0.0
    try:
        eng = Engine(cfg)
    except Exception as ex:
        print(f"{name:34s} path={path} CREATE FAILED: {ex}")
        return
    eng.set_params(flat)
    xt = torch.as_tensor(x, device="cuda")
    eng.bind_points(xt)
    if x_bc is not None:
        eng.bind_boundary(torch.as_tensor(x_bc, device="cuda"))
    jets = eng.forward_jets(xt).cpu().numpy()
    ej = [relerr(jets[c], ojets[c]) for c in range(jets.shape[0])]
    sc, psi, res = eng.residual()
    sc2 = eng.step()
    grad = eng.get_grad()
    eg = relerr(grad, ograd)
    # per-layer gradient error
    offs = []
    o = 0
    for i in range(len(layers) - 1):
        nW = layers[i] * layers[i + 1]
        offs.append((f"W{i}", o, o + nW)); o += nW
        offs.append((f"b{i}", o, o + layers[i + 1])); o += layers[i + 1]
    worst = max(offs, key=lambda t: np.abs(grad[t[1]:t[2]] - ograd[t[1]:t[2]]).max() / (np.abs(ograd).max() + 1e-300))
    print(f"{name:34s} path={eng.active_path} jets {max(ej):.1e} mu {sc['mu']:.6g}/{osc['mu']:.6g} "
          f"loss {sc2['loss']:.6g}/{osc['loss']:.6g} pde {relerr(sc2['pde'], osc['pde']):.1e} bc {sc2['bc']:.3g}/{osc['bc']:.3g} "
          f"sym {sc2['sym']:.3g}/{osc['sym']:.3g} psi {relerr(psi.cpu().numpy(), ores['psi']):.1e} "
          f"res {relerr(res.cpu().numpy(), ores['residual']):.1e} GRAD {eg:.1e} (worst {worst[0]}) gn {sc2['grad_norm']:.5g}/{np.linalg.norm(ograd):.5g}",
          flush=True)
    eng.close()


if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0), flush=True)
    cases = [
        ("1d_64x3_refine_m0", dict(layers=[1, 64, 64, 64, 1], activation=1, kinetic_coeff=1.0, pot_scale=1.0, gamma=5.0, p=3,
                                   base_mode=0, perturb_scale=0.05, dx=12 / 499), 500),
        ("1d_32x4_nb_sym", dict(layers=[1, 32, 32, 32, 32, 1], gamma=1.0, p=3, base_mode=0, base_deriv=1, w_sym=5.0,
                                dx=12 / 299), 300),
        ("1d_64x4_m2_p4", dict(layers=[1, 64, 64, 64, 64, 1], activation=1, gamma=2.0, p=4, base_mode=2, perturb_scale=0.1,
                               dx=12 / 1000), 1001),
        ("2d_64x4_g500", dict(layers=[2, 64, 64, 64, 64, 1], gamma=500.0, dx=36 / 777), 777),
        ("2d_32x3", dict(layers=[2, 32, 32, 32, 1], gamma=10.0, dx=0.05), 100),
        ("3d_64x3", dict(layers=[3, 64, 64, 64, 1], gamma=20.0, dx=0.01, omega=(1.0, 1.4, 2.0)), 130),
        ("2d_64x3_complex_rot", dict(layers=[2, 64, 64, 64, 2], complex_psi=True, gamma=30.0, dx=0.02, omega_rot=0.8), 200),
        ("2d_tiny_N5", dict(layers=[2, 64, 64, 1], gamma=3.0, dx=0.1), 5),
    ]
    for name, kw, N in cases:
        for path in (gpe_pinn.PATH_GENERIC, gpe_pinn.PATH_FUSED):
            run_case(name, kw, N, path)
    gen_only = [
        ("2d_128x3", dict(layers=[2, 128, 128, 128, 1], gamma=100.0, dx=0.01), 300),
        ("2d_100x2_odd", dict(layers=[2, 100, 100, 1], gamma=1.0, dx=0.01), 77),
        ("3d_256x2", dict(layers=[3, 256, 256, 1], gamma=100.0, dx=0.01), 64),
        ("1d_64_single_hidden", dict(layers=[1, 64, 1], gamma=1.0, dx=0.01, base_mode=1), 50),
    ]
    for name, kw, N in gen_only:
        run_case(name, kw, N, gpe_pinn.PATH_AUTO, scale=0.15)
