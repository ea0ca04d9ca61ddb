#!/usr/bin/env python3
"""Randomised parity sweep: engine (kernel set as gpe_create picks it: fused / wide / padded / residual / generic) against the fp64 oracle on
random problem descriptions -- dimensions, hidden widths (native, odd, ragged), depth, activation, residual blocks, loss terms (Riesz forms,
energy-functional lambda, regularisers, symmetry), batch sizes down to one point.  usage: python tools/fuzz_parity.py [cases] [seed] [smallest N] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gpe_pinn
from oracle import gpe_oracle as go
from tests.test_gpu_parity import cfg_from_problem
from tests import helpers as H

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # > 1: also follow the fp64 oracle's optimiser for that many steps (loss per step)
dp_mode = len(sys.argv) > 5 and sys.argv[5] == "dp"             # also: the step of TWO engines on the two halves of the points (world_size = 2, exchange
                                                                # buffers summed by hand as the all-reduces would) against the single engine's
bad = 0
t0 = time.time()
for it in range(cases):
    d = int(rng.choice([1, 2, 3], p=[0.4, 0.4, 0.2]))
    residual = rng.random() < 0.15
    act = int(rng.random() < 0.4)
    if residual:
        Hh = int(rng.choice([32, 64, 128]))
        layers = [d] + [Hh] * int(rng.integers(2, 5)) + [1]
    else:
        depth = int(rng.integers(2, 7))
        if rng.random() < 0.25:
            layers = [d] + [int(rng.choice([16, 20, 32, 48, 64, 100])) for _ in range(depth)] + [1]
        else:
            layers = [d] + [int(rng.choice([20, 32, 48, 64, 100, 128, 200, 256, 300]))] * depth + [1]
    cplx = (not residual) and d >= 2 and rng.random() < 0.12
    if cplx:
        layers[-1] = 2
    kw = dict(layers=layers, activation=act, net_kind=go.NET_RESIDUAL if residual else go.NET_MLP, gamma=float(rng.choice([0.0, 1.0, 20.0, 200.0])),
              kinetic_coeff=float(rng.choice([0.5, 1.0])), dx=float(rng.choice([0.01, 0.05])), complex_psi=bool(cplx))
    if cplx:
        kw.update(omega_rot=float(rng.choice([0.0, 0.8])))
    else:
        kw.update(p=int(rng.choice([3, 3, 5, 2, 4])), abs_power=bool(rng.random() < 0.5))
        flav = rng.random()
        if flav < 0.2:
            kw.update(w_riesz=float(rng.choice([0.05, 1.0])), riesz_kind=int(rng.integers(0, 3)))
        elif flav < 0.4 and kw["p"] % 2 == 1:
            kw.update(lambda_kind=go.LAMBDA_ENERGY, w_reg_f=float(rng.choice([0.0, 1.0])), w_reg_lam=float(rng.choice([0.0, 1.0])), w_norm=float(rng.choice([0.0, 20.0])))
        elif flav < 0.5:
            kw.update(w_reg_f=0.5)
        if d == 1 and rng.random() < 0.5:
            kw.update(base_mode=int(rng.integers(0, 4)), perturb_scale=float(rng.choice([1.0, 0.05])))
            if rng.random() < 0.3:          # box base + hard boundary factor (refine/box_pinn_simulation.py)
                kw.update(base_kind=go.BASE_BOX, envelope=go.ENV_SIN, box_L=10.0, env_L=10.0)
        if d == 1 and rng.random() < 0.4:
            kw.update(potential=int(rng.choice([go.POT_GAUSSIAN, go.POT_PERIODIC, go.POT_NONE])), pot_a=0.5)
        if rng.random() < 0.15:
            kw.update(w_sym=5.0, sym_sign=float(rng.choice([1.0, -1.0])))
    if d == 3:
        kw.update(omega=(1.0, 1.4, 2.0))
    N = max(min_n, int(rng.choice([1, 7, 16, 17, 100, 333, 1000, 3000])))
    if N < 4:                # one point: lambda = u Hu / u^2 makes the residual identically zero -- fp32 computes round-off times gamma p u^(p-1) there
        kw["gamma"] = min(kw["gamma"], 1.0)
        pb = None
    wmax = max(layers[1:-1])
    scale = 0.3 if wmax <= 64 else (0.15 if wmax <= 128 else 0.08)
    x = (np.linspace(-5, 5, N).reshape(-1, 1) if d == 1 else rng.uniform(-3, 3, (N, d))).astype(np.float32)
    xb = (np.array([[-5.0], [5.0]]) if d == 1 else rng.uniform(-3, 3, (5, d))).astype(np.float32)
    flat = (rng.normal(0, 1, go.param_count(layers, kw["net_kind"])) * scale).astype(np.float32)
    pb = go.Problem(**kw)
    only = os.environ.get("FUZZ_ONLY")
    if only is not None and int(only) != it:
        continue
    if only is not None:          # one case in detail: per-step losses of the oracle, the engine as picked, and the generic set on the network as given
        st = go.OptState(lr0=1e-3)
        _, tr = go.train_steps(pb, st, flat.astype(np.float64), x.astype(np.float64), n_steps, xb.astype(np.float64), dtype=np.float64)
        rows = {"oracle": [t["loss"] for t in tr]}
        for name, over in (("auto", {}), ("generic", dict(path=gpe_pinn.PATH_GENERIC))):
            eng = gpe_pinn.Engine(cfg_from_problem(pb, **over))
            eng.set_params(flat); eng.bind_points(torch.as_tensor(x, device="cuda")); eng.bind_boundary(torch.as_tensor(xb, device="cuda"))
            rows[name] = [eng.step()["loss"] for _ in range(n_steps)]
            rows[name + "_gn"] = eng.read_scalars()["grad_norm"]
            eng.close()
        print(layers, kw)
        for k in range(n_steps):
            print(k, " ".join(f"{nm} {rows[nm][k]:.9g}" for nm in ("oracle", "auto", "generic")))
        print("grad norms", rows["auto_gn"], rows["generic_gn"], "oracle gn", [t["grad_norm"] for t in tr])
        sys.exit(0)
    try:
        osc, ograd, _ = go.full_loss_and_grad(pb, flat.astype(np.float64), x.astype(np.float64), xb.astype(np.float64))
        eng = gpe_pinn.Engine(cfg_from_problem(pb))
        eng.set_params(flat)
        eng.bind_points(torch.as_tensor(x, device="cuda"))
        eng.bind_boundary(torch.as_tensor(xb, device="cuda"))
        kern = eng.active_kernels
        sc = eng.step()
        g = eng.get_grad()
        traj = 0.0
        if n_steps > 1:
            st = go.OptState(lr0=1e-3)
            _, tr = go.train_steps(pb, st, flat.astype(np.float64), x.astype(np.float64), n_steps, xb.astype(np.float64), dtype=np.float64)
            losses = [sc["loss"]] + [eng.step()["loss"] for _ in range(n_steps - 1)]
            traj = max(abs(a - t["loss"]) / max(abs(t["loss"]), 1e-30) / (1 + k) for k, (a, t) in enumerate(zip(losses, tr)))
        eng.close()
        dpe = 0.0
        if dp_mode and N >= 2:
            import dataclasses
            pbn = dataclasses.replace(pb, n_global=N)
            lo = N // 2
            engs = []
            for xs in (x[:lo], x[lo:]):
                e2 = gpe_pinn.Engine(cfg_from_problem(pbn, world_size=2))
                e2.set_params(flat); e2.bind_points(torch.as_tensor(xs, device="cuda")); e2.bind_boundary(torch.as_tensor(xb, device="cuda"))
                engs.append(e2)
            for e2 in engs: e2.step_begin()
            tot = engs[0].exchange_sums + engs[1].exchange_sums
            for e2 in engs:
                e2.exchange_sums.copy_(tot); e2.step_backward()
            gt = engs[0].exchange_grad + engs[1].exchange_grad
            for e2 in engs:
                e2.exchange_grad.copy_(gt); e2.step_update()
            s2 = engs[0].read_scalars()
            dpe = max(abs(s2["loss"] - sc["loss"]) / max(abs(sc["loss"]), 1e-30), abs(s2["mu"] - sc["mu"]) / max(abs(sc["mu"]), 1e-6),
                      H.rel_err(engs[0].get_grad(), g) / 5.0)
            same = np.array_equal(engs[0].get_params(), engs[1].get_params())
            for e2 in engs: e2.close()
            if not same: dpe = float("inf")
        traj = max(traj, 50.0 * dpe)          # (reported in the same column: the two-rank step within 2e-5 of the single engine's, replicas bit-identical)
        f = 10.0 if N < 4 else 1.0
        el = abs(sc["loss"] - osc["loss"]) / max(abs(osc["loss"]), 1e-30)
        em = abs(sc["mu"] - osc["mu"]) / max(abs(osc["mu"]), 1e-6)
        eg = H.rel_err(g, ograd)
        ok = el < f * 2e-4 and em < f * 5e-5 and eg < f * 1e-4 and traj < f * 1e-3 and np.isfinite(el + em + eg + traj)
        tag = "ok " if ok else "BAD"
    except Exception as ex:           # noqa: BLE001 -- the sweep reports, it does not stop
        ok, tag, el, em, eg, traj, kern = False, "EXC", float("nan"), float("nan"), float("nan"), float("nan"), {"fwd": str(ex)[:80], "bwd": ""}
    bad += 0 if ok else 1
    if not ok or it % 10 == 0:
        print(f"{tag} #{it:3d} N={N:5d} {layers} act={act} res={int(residual)} {{{', '.join(f'{k}={v}' for k, v in kw.items() if k not in ('layers', 'activation', 'net_kind', 'dx', 'kinetic_coeff'))}}} "
              f"loss {el:.1e} mu {em:.1e} grad {eg:.1e} traj {traj:.1e}  {kern['fwd'][:34]} / {kern['bwd'][:30]}", flush=True)
print(f"{cases} cases, {bad} failures, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
