#!/usr/bin/env python3
"""bench.py -- throughput of the GPE eigenvalue-residual training step on MI355X.

A "step" is one full pass of the hot path over the collocation batch: jet forward (psi, grad psi, Laplacian),
Rayleigh quotient mu, residual, boundary + normalisation penalties, reverse pass, grad-norm clip, Adam, scheduler.

Workload (config.workload = "ns_2d_4x64"): BASELINE.json's north-star configuration -- the one its metric target is
quoted on -- 2D isotropic harmonic trap, g = 500, MLP [2,64,64,64,64,1] (4 hidden x 64), 1 048 576 collocation points
per GPU (1024 x 1024 uniform grid per rank), 512 boundary points; synthetic seeded weights (reference init a13).

--gpus N (N > 1): when RANK is not in the environment this process is only a launcher -- it starts N fresh worker processes
(one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE anything touches the GPU and relays their exit codes;
under torchrun (RANK set) it is a worker.  Workers shard the points and exchange through the ENGINE's own RCCL communicator
(gpe_comm_init / gpe_step_dp: two all-reduces per step on a dedicated HIP stream, no Python between the phases);
torch.distributed is used for the rendezvous of the ncclUniqueId, the barriers and the max-over-ranks of the time.
--scaling weak (default): every rank owns a full per-GPU grid.  --scaling strong: BASELINE's global size is split over ranks.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      -- dominant kernel (jet reverse pass) algorithmic FLOP / HIP-event time vs the fp32 MFMA peak
  cpu_baseline  -- the reference's op sequence (torch-autograd restatement, oracle/torch_ref.py) and the native C++/OpenMP
                   jet restatement (oracle/cpu_ref) timed on this host
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PROFILE_ROUND, PROFILE_ROUND_PREV = "r04", "r03"
PROFILE_DIR = os.path.join(ROOT, "profiles", PROFILE_ROUND)
PROFILE_DIR_PREV = os.path.join(ROOT, "profiles", PROFILE_ROUND_PREV)

WORKLOADS = {
    # name: layers, per-GPU grid (weak scaling), global grid of the BASELINE config (strong scaling), gamma, domain half-width
    "ns_2d_4x64": dict(layers=[2, 64, 64, 64, 64, 1], grid=(1024, 1024), global_grid=(1024, 1024), gamma=500.0, half=8.0),
    "cfg1_1d_4x32": dict(layers=[1, 32, 32, 32, 32, 1], grid=(2048,), global_grid=(2048,), gamma=0.0, half=10.0),
    "cfg2_1d_4x64": dict(layers=[1, 64, 64, 64, 64, 1], grid=(65536,), global_grid=(65536,), gamma=100.0, half=10.0),
    # BASELINE configs[2]: 1 048 576 points over 8 GPUs = 131 072 per GPU
    "cfg3_2d_5x128": dict(layers=[2, 128, 128, 128, 128, 128, 1], grid=(512, 256), global_grid=(1024, 1024), gamma=500.0, half=8.0),
    # BASELINE configs[3]: rotating trap, complex psi (n_out = 2), 6x128; 2 097 152 points over 8 GPUs = 262 144 per GPU
    "cfg4_2d_6x128_rot": dict(layers=[2, 128, 128, 128, 128, 128, 128, 2], grid=(512, 512), global_grid=(2048, 1024), gamma=500.0,
                              half=8.0, complex_psi=True, omega_rot=0.8),
    # BASELINE configs[4]: 3D anisotropic trap omega = (1, 1.4, 2), g = 1000, 6x256; 4 194 304 points over 8 GPUs = 524 288 per GPU
    "cfg5_3d_6x256": dict(layers=[3, 256, 256, 256, 256, 256, 256, 1], grid=(64, 128, 64), global_grid=(256, 128, 128), gamma=1000.0,
                          half=6.0, omega=(1.0, 1.4, 2.0), generic_ok=True),
}


def reference_init(layers, seed=0, mode=0):
    """advanced_initialization of the notebook surface (nb c18): Xavier-uniform(gain 1/(1+0.1 mode)), bias 0.01."""
    torch.manual_seed(seed)
    flat = []
    for i in range(len(layers) - 1):
        W = torch.empty(layers[i + 1], layers[i])
        torch.nn.init.xavier_uniform_(W, gain=1.0 / (1.0 + 0.1 * mode))
        flat.append(W.reshape(-1))
        flat.append(torch.full((layers[i + 1],), 0.01))
    return torch.cat(flat).numpy().astype(np.float32)


def make_points(wl, rank, world, scaling="weak"):
    """Shard `rank` of the point grid: contiguous block along the first axis (grid order keeps the quadrature additive).
    weak: the global grid is `world` per-GPU grids side by side; strong: the BASELINE global grid cut into `world` blocks."""
    half = wl["half"]
    if scaling == "strong":
        g = tuple(wl["global_grid"])
        if g[0] % world:
            raise SystemExit(f"strong scaling: first grid axis {g[0]} is not divisible by {world} ranks")
        n0, n0_glob = g[0] // world, g[0]
    else:
        g = tuple(wl["grid"])
        n0, n0_glob = g[0], g[0] * world
    x0 = np.linspace(-half, half, n0_glob, dtype=np.float64)
    h0 = 2 * half / (n0_glob - 1)
    xs = x0[rank * n0:(rank + 1) * n0]
    if len(g) == 1:
        x = xs.reshape(-1, 1)
        dx = h0
        xb = np.array([[-half], [half]])
    elif len(g) == 3:
        ys = np.linspace(-half, half, g[1], dtype=np.float64)
        zs = np.linspace(-half, half, g[2], dtype=np.float64)
        X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        dx = h0 * (2 * half / (g[1] - 1)) * (2 * half / (g[2] - 1))
        t = np.linspace(-half, half, 23, endpoint=False)
        A, B = np.meshgrid(t, t, indexing="ij")
        xb = np.stack([A.ravel(), B.ravel(), np.full(A.size, half)], axis=1)          # one face of the box, 529 points
    else:
        ys = np.linspace(-half, half, g[1], dtype=np.float64)
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel()], axis=1)
        dx = h0 * (2 * half / (g[1] - 1))
        t = np.linspace(-half, half, 128, endpoint=False)
        xb = np.concatenate([np.stack([t, np.full_like(t, -half)], 1), np.stack([np.full_like(t, half), t], 1),
                             np.stack([-t, np.full_like(t, half)], 1), np.stack([np.full_like(t, -half), -t], 1)])
    return x.astype(np.float32), float(dx), xb.astype(np.float32)


def oracle_problem(wl, dx):
    from oracle import gpe_oracle as go
    return go.Problem(layers=wl["layers"], gamma=wl["gamma"], p=3, kinetic_coeff=0.5, pot_scale=0.5, dx=dx,
                      w_bc=10.0, w_norm=20.0, complex_psi=bool(wl.get("complex_psi", False)),
                      omega_rot=float(wl.get("omega_rot", 0.0)), omega=tuple(wl.get("omega", (1.0, 1.0, 1.0))))


def cpu_baseline(wl, flat, budget_s=12.0, n_sample=8192):
    """The reference's own op sequence on the host cores (bounded sample of the same workload)."""
    from oracle import gpe_oracle as go
    from oracle import torch_ref as tr
    x, dx, xb = make_points(wl, 0, 1)
    idx = np.linspace(0, x.shape[0] - 1, n_sample).astype(np.int64)
    xs = x[idx]
    pb = oracle_problem(wl, dx)
    trn = tr.TorchTrainer(pb, flat, xs, xb, lr=1e-3, sched=go.SCHED_CONST)
    ncpu = os.cpu_count() or 1
    best = None
    sweep = sorted({1, min(8, ncpu), min(32, ncpu)})
    for nt in sweep:                                   # thread sweep: small-tensor autograd rarely scales with cores
        torch.set_num_threads(nt)
        trn.step()                                     # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            trn.step()
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / len(sweep) or n >= 200:
                break
        rate = n * n_sample / el
        if best is None or rate > best[0]:
            best = (rate, nt, n, el)
    rate, nt, n, el = best
    return dict(value=rate, unit="points/s", cores=int(nt), kind="port", host_cpus=int(ncpu),
                sample=f"{n} full training steps of the torch-autograd restatement (oracle/torch_ref.py, the reference's "
                       f"op sequence) on a {n_sample}-point slice of the same workload, fp32, {el:.1f} s; best of a "
                       f"thread sweep {sweep}")


def cpu_baseline_native(wl, flat, budget_s=12.0, n_sample=262144):
    """The like-for-like algorithm on the host: oracle/cpu_ref (C++/OpenMP forward-mode jets + hand-derived reverse pass, fp32),
    thread sweep {1, all cores}.  Checker-side code (oracle/): timed here, never shipped."""
    try:
        from oracle import cpu_ref
        lib = cpu_ref.load()
    except Exception as ex:                                  # not built on this host: report, do not fail the bench
        return dict(value=None, kind="native", error=str(ex)[:200])
    x, dx, xb = make_points(wl, 0, 1)
    idx = np.linspace(0, x.shape[0] - 1, n_sample).astype(np.int64)
    xs = np.ascontiguousarray(x[idx])
    pb = oracle_problem(wl, dx)
    ncpu = os.cpu_count() or 1
    out = {}
    sweep = sorted({1, min(16, ncpu), min(64, ncpu), ncpu})
    for nt in sweep:
        ns = n_sample if nt > 1 else n_sample // 8           # one thread: a shorter slice of the same points
        xt = xs[:ns]
        cpu_ref.step(lib, pb, flat, xt, threads=nt)          # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            cpu_ref.step(lib, pb, flat, xt, threads=nt)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / len(sweep) or n >= 400:
                break
        out[nt] = n * ns / el
    best_nt = max(out, key=out.get)
    return dict(value=out[best_nt], unit="points/s", cores=int(best_nt), kind="native", host_cpus=int(ncpu),
                per_threads={str(k): v for k, v in out.items()},
                sample=f"forward jets + residual + reverse pass (loss and gradient, no Adam) of oracle/cpu_ref (C++/OpenMP, fp32) "
                       f"on a {n_sample}-point slice of the same workload; thread sweep {sorted(out)}")


def launch_workers(args):
    """--gpus N without RANK in the environment: start N fresh worker processes.  Nothing here touches the GPU
    (a process that has initialised HIP must not be replaced or forked on this pool)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # Poll the ranks: a rank that dies (out of memory, RCCL init) leaves the others blocked in a collective for ever, so the first
    # non-zero exit -- or the overall time limit -- ends the siblings (fresh child processes: plain terminate / kill) and the launcher
    # exits non-zero naming the rank.
    deadline = time.monotonic() + float(os.environ.get("GPE_BENCH_LAUNCH_TIMEOUT", args.launch_timeout))
    failed, timed_out = None, False
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(rc == 0 for rc in rcs):
            sys.exit(0)
        if time.monotonic() > deadline:
            timed_out = True
            break
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.monotonic() + 10.0
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    if timed_out:
        print(f"bench launcher: time limit reached, {sum(rc is None for rc in rcs)} of {args.gpus} ranks still running; all terminated",
              file=sys.stderr, flush=True)
        sys.exit(124)
    print(f"bench launcher: rank {failed[0]} exited with code {failed[1]}; the other ranks were terminated", file=sys.stderr, flush=True)
    sys.exit(abs(failed[1]) if abs(failed[1]) < 256 else 1)


def load_profile_json(name):
    """-> (record, path, stale): this round's committed record, else the previous round's -- then `stale` names that round, and the
    line labels every figure taken from it as coming from an earlier build."""
    for d, rnd in ((PROFILE_DIR, None), (PROFILE_DIR_PREV, PROFILE_ROUND_PREV)):
        p = os.path.join(d, name)
        if os.path.exists(p):
            try:
                return json.load(open(p)), os.path.relpath(p, ROOT), rnd
            except Exception:
                pass
    return None, None, None


def split_bf16_mode(cfg, flat, x, xb, local_rank, args, n_local):
    """The same workload, same steps, on the opt-in kernels that form every H x H product from six bf16 matrix products on three-piece
    splits of both operands (GPE_FWD_B6=1 GPE_BWD_B6=1; fp32-equivalent results: DESIGN.md section 4, tests
    test_split_bf16_kernels_match_oracle).  Reported next to the headline value, which stays on the fp32 matrix instruction."""
    import gpe_pinn
    os.environ["GPE_FWD_B6"] = "1"
    os.environ["GPE_BWD_B6"] = "1"
    try:
        eng = gpe_pinn.Engine(cfg, device=local_rank)
    finally:
        os.environ.pop("GPE_FWD_B6", None)
        os.environ.pop("GPE_BWD_B6", None)
    try:
        eng.set_params(flat)
        eng.bind_points(torch.as_tensor(x, device=f"cuda:{local_rank}"))
        eng.bind_boundary(torch.as_tensor(xb, device=f"cuda:{local_rank}"))
        kernels = eng.active_kernels
        eng.run(args.warmup)
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sc = eng.read_scalars()
    finally:
        eng.close()
    return {"switches": "GPE_FWD_B6=1 GPE_BWD_B6=1", "kernels": kernels, "value": n_local * args.steps / dt, "unit": "points/s",
            "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "loss_after_steps": sc["loss"],
            "note": "six v_mfma_f32_16x16x32_bf16 per fp32 product on three bf16 pieces per operand, fp32 accumulation; forward maps and "
                    "the reverse pass's adjoint products (the weight-gradient products stay on v_mfma_f32_16x16x4_f32); the board runs "
                    "about 9 % lower clocks in this mode (power limit)"}


def parity_points(layers, n_local, budget_s):
    """Points the in-run oracle check covers: the whole bound batch (the launch configuration that was TIMED: uneven tile split,
    head inside the forward kernel) when the numpy oracle fits the time budget, else the largest slice that does (>= 4096).  Cost
    model measured on the GPU box's host: ~4.5e-9 s per (parameter x point) for the two-phase sharded oracle."""
    P = sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))
    n_fit = int(budget_s / (4.5e-9 * P))
    if n_fit >= n_local:
        return n_local
    return max(4096, min(n_local, n_fit // 4096 * 4096))


def parity_check(eng, wl, flat, x, dx, xb, n_check, world_pts, timed_kernels):
    """Correctness evidence measured in this process, after the timed region, on the bench engine itself: its parameters and
    optimiser are reset, ONE step runs on the bound batch AS TIMED (or, for the workloads whose oracle would take minutes, on a slice
    of the same points), and loss, mu and the gradient are compared with the fp64 oracle (oracle/gpe_oracle.py -- the checker, never
    the thing measured; sharded two-phase evaluation over 65 536-point chunks)."""
    from oracle import gpe_oracle as go
    full = n_check >= x.shape[0]
    if full:
        xs = x
    else:
        idx = np.linspace(0, x.shape[0] - 1, n_check).astype(np.int64)
        xs = np.ascontiguousarray(x[idx])
    pb = oracle_problem(wl, dx)
    t0 = time.perf_counter()
    osc, ograd = go.sharded_loss_and_grad(pb, flat.astype(np.float64), xs.astype(np.float64), xb.astype(np.float64),
                                          chunk=65536, threads=min(8, os.cpu_count() or 1))
    t_or = time.perf_counter() - t0
    eng.set_params(flat)
    eng.reset_optimizer(1e-3)
    if not full:
        eng.bind_points(torch.as_tensor(xs, device=f"cuda:{eng.device}"))
        eng.set_n_global(n_check)
    kernels = eng.active_kernels
    sc = eng.step()
    g = eng.get_grad()
    gerr = float(np.abs(g - ograd).max() / np.abs(ograd).max())
    res = {"points": int(xs.shape[0]), "batch": "the bound batch as timed" if full else f"{n_check}-point slice of the timed batch",
           "oracle": "oracle/gpe_oracle.py (numpy fp64, sharded two-phase evaluation), same seeded weights, same points",
           "kernels": kernels, "timed_kernels": timed_kernels, "same_kernels_as_timed": kernels == timed_kernels,
           "uneven_split": kernels.get("split", "") not in ("", "fwd 0/1024, bwd 0/1024"),
           "loss": sc["loss"], "loss_oracle": float(osc["loss"]), "loss_rel_err": abs(sc["loss"] - osc["loss"]) / abs(osc["loss"]),
           "mu": sc["mu"], "mu_oracle": float(osc["mu"]), "mu_rel_err": abs(sc["mu"] - osc["mu"]) / abs(osc["mu"]),
           "grad_max_err_over_max_abs": gerr, "tolerance": {"loss_rel": 1e-4, "mu_rel": 2e-5, "grad": 5e-5}, "oracle_seconds": t_or}
    res["ok"] = bool(res["loss_rel_err"] < 1e-4 and res["mu_rel_err"] < 2e-5 and gerr < 5e-5)
    if not full:
        eng.set_n_global(world_pts)
    return res


def dp_crosscheck(eng, flat, lr=1e-3):
    """N > 1, --exchange engine: before anything is timed, one step with the engine's own RCCL exchange and one with the
    torch.distributed protocol from the same state; loss, mu and the all-reduced gradient must agree.  -> (ok, detail)"""
    eng.set_params(flat)
    eng.reset_optimizer(lr)
    eng.run_dp(1)
    eng.synchronize()
    a, ga = eng.read_scalars(), eng.get_grad()
    eng.set_params(flat)
    eng.reset_optimizer(lr)
    eng.step_distributed()
    eng.synchronize()
    b, gb = eng.read_scalars(), eng.get_grad()
    eng.set_params(flat)
    eng.reset_optimizer(lr)
    gerr = float(np.abs(ga - gb).max() / max(np.abs(gb).max(), 1e-30))
    lerr = abs(a["loss"] - b["loss"]) / max(abs(b["loss"]), 1e-30)
    merr = abs(a["mu"] - b["mu"]) / max(abs(b["mu"]), 1e-30)
    ok = bool(np.isfinite(gerr) and gerr < 1e-5 and lerr < 1e-5 and merr < 1e-6)
    return ok, {"grad_rel": gerr, "loss_rel": lerr, "mu_rel": merr}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ns_2d_4x64", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="default: weak (every rank a full per-GPU grid); cfg3 / cfg4 / cfg5 at --gpus 8 default to strong, i.e. to "
                         "BASELINE's global sizes")
    ap.add_argument("--blocks", type=int, default=50,
                    help="the timed K-step block is repeated this many times (each bracketed by barrier + synchronize); the line "
                         "reports the median block, min and max.  Shortened automatically when 50 blocks would exceed ~20 s")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher (--gpus N): overall limit in seconds")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--parity-points", type=int, default=0,
                    help="points of the in-run oracle check; 0 = the whole timed batch when the oracle fits --parity-budget, else a slice")
    ap.add_argument("--parity-budget", type=float, default=90.0, help="seconds of host time the in-run oracle check may take")
    ap.add_argument("--exchange", default="engine", choices=["engine", "torch"],
                    help="N > 1: all-reduces issued by the engine (native RCCL) or by torch.distributed between the phases")
    ap.add_argument("--async-grad", action="store_true",
                    help="OPT-IN one-step-stale gradient: the gradient all-reduce overlaps the next forward (changes the trajectory; labelled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the second timed pass on the split-bf16 kernels (H <= 64 workloads)")
    args = ap.parse_args()

    if args.scaling is None:
        big = args.workload in ("cfg3_2d_5x128", "cfg4_2d_6x128_rot", "cfg5_3d_6x256")
        args.scaling = "strong" if (big and int(os.environ.get("WORLD_SIZE", args.gpus)) == 8) else "weak"
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_workers(args)                               # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("GPE_BENCH_LAUNCH_TEST"):            # CPU test of the launcher: report the rank environment, touch nothing
        print(json.dumps({"launch_test": True, "rank": rank, "local_rank": local_rank, "world": world,
                          "master": os.environ.get("MASTER_ADDR"), "gpus_arg": args.gpus, "scaling": args.scaling}), flush=True)
        mode = os.environ["GPE_BENCH_LAUNCH_TEST"]          # "fail:<rank>": that rank exits 3, the others hang like ranks in a collective
        if mode.startswith("fail:"):
            if rank == int(mode.split(":")[1]):
                sys.exit(3)
            time.sleep(600)
        if mode == "hang":
            time.sleep(600)
        return
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun always take the RCCL path, even at world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()

    import gpe_pinn
    wl = WORKLOADS[args.workload]
    layers = wl["layers"]
    flat = reference_init(layers, seed=0)
    x, dx, xb = make_points(wl, rank, world, args.scaling)
    n_local = x.shape[0]
    cfg = gpe_pinn.GPEConfig(layers=layers, gamma=wl["gamma"], p=3, kinetic_coeff=0.5, pot_scale=0.5, dx=dx,
                             w_bc=10.0, w_norm=20.0, lr=1e-3, n_global=n_local * world, world_size=world,
                             omega=tuple(wl.get("omega", (1.0, 1.0, 1.0))),
                             complex_psi=bool(wl.get("complex_psi", False)), omega_rot=float(wl.get("omega_rot", 0.0)))
    eng = gpe_pinn.Engine(cfg, device=local_rank)
    fused = eng.active_path == gpe_pinn.PATH_FUSED
    if not fused and not wl.get("generic_ok", False):
        raise SystemExit("bench: fused MFMA path not active")
    eng.set_params(flat)
    eng.bind_points(torch.as_tensor(x, device=f"cuda:{local_rank}"))      # inputs resident in HBM before timing
    eng.bind_boundary(torch.as_tensor(xb, device=f"cuda:{local_rank}"))
    kernels = eng.active_kernels
    native = use_dist and args.exchange == "engine"
    xcheck = None
    if native:
        # a rank whose communicator cannot be created (librccl missing / ncclCommInitRank error) must not take the job down: every rank
        # learns of it through the torch.distributed group and the whole job drops to the torch protocol, labelled in the line
        try:
            eng.comm_init(rank, world)
            bad = 0
        except Exception as ex:                                        # noqa: BLE001 -- reported, then agreed on collectively
            print(f"bench: rank {rank}: engine RCCL communicator not created ({ex}); falling back to torch.distributed", file=sys.stderr, flush=True)
            bad = 1
        flag = torch.tensor([bad], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) != 0:
            native = False
            xcheck = {"ok": False, "comm_init_failed": True}
    if native:
        if world > 1 or os.environ.get("GPE_BENCH_FORCE_XCHECK"):     # (the variable: run the check at world 1 too -- tests)
            # the engine-native exchange against the torch.distributed protocol, from the same state, before anything is timed: a
            # mismatch on any rank sends the whole job to the torch protocol (labelled in the line) instead of timing a wrong path
            ok, detail = dp_crosscheck(eng, flat)
            flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=f"cuda:{local_rank}")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            xcheck = dict(detail, ok=bool(flag.item() == 0))
            if not xcheck["ok"]:
                native = False
                if rank == 0:
                    print(f"bench: engine RCCL exchange disagrees with the torch.distributed protocol {detail}; using the latter",
                          file=sys.stderr, flush=True)
        if native and args.async_grad:
            eng.comm_set_async(True)

    def run_steps(k):
        if not use_dist:
            eng.run(k)
        elif native:
            eng.run_dp(k)
        else:
            for _ in range(k):
                eng.step_distributed()

    coll0 = eng.comm_info()["collectives"] if native else 0          # (the cross-check above issued some)
    run_steps(args.warmup)
    eng.synchronize()
    eng.profile_enable(True)           # HIP events around the two dominant kernels, on the engine's stream, over the timed region

    def timed_block():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; the maximum over the ranks."""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # One K-step block is a few tens of milliseconds -- a single scheduling hiccup moves it by per cent and an outside observer
    # never sees the GPU busy.  The block is therefore repeated (>= 3 s of GPU time for the default workload) and the line reports
    # the MEDIAN block next to min and max; every block is a contract-conforming measurement of exactly K steps.
    block_s = [timed_block()]
    n_blocks = max(1, min(args.blocks, max(5, int(20.0 / max(block_s[0], 1e-6)))))
    for _ in range(n_blocks - 1):
        block_s.append(timed_block())
    elapsed = float(np.median(block_s))
    prof = eng.profile_read()
    eng.profile_enable(False)
    sc = eng.read_scalars()
    comm = eng.comm_info()
    steps_from_init = args.warmup + args.steps * len(block_s)

    if rank == 0:
        flops_pt, bmat_pt = eng.step_cost()
        f_fwd = flops_pt / 3.0
        # channels the training batches carry: value, d first derivatives, ONE Laplacian channel (SURVEY 8(d) counts 1 + 2d;
        # its per-point figure is reported next to the executed one -- the roofline fraction uses only executed flops)
        d_in, Hh, Lh = layers[0], layers[1], len(layers) - 2
        chan = d_in + 2
        gemm_fwd = 2.0 * d_in * Hh + chan * (2.0 * Hh * Hh * (Lh - 1) + 2.0 * Hh * layers[-1])
        survey_fwd = 2.0 * d_in * Hh + (1 + 2 * d_in) * (2.0 * Hh * Hh * (Lh - 1) + 2.0 * Hh * layers[-1]) + Hh * Lh * (3 + 5 * d_in)
        pts = n_local * world * args.steps
        value = pts / elapsed
        n_rows = n_local + (int(xb.shape[0]) if xb.shape[0] * 8 <= n_local else 0)     # merged boundary rows ride in the launch
        bwd_s = prof["bwd_ms"] / max(1, prof["bwd_launches"]) * 1e-3
        fwd_s = prof["fwd_ms"] / max(1, prof["fwd_launches"]) * 1e-3
        ach = 2.0 * f_fwd * n_rows / bwd_s / 1e12 if bwd_s > 0 else 0.0
        ach_gemm = 2.0 * gemm_fwd * n_rows / bwd_s / 1e12 if bwd_s > 0 else 0.0
        ach_f = f_fwd * n_rows / fwd_s / 1e12 if fwd_s > 0 else 0.0
        # HBM bytes per launch of the dominant kernel and the SQ counter shares: from the committed rocprofv3 --pmc passes of
        # this same command (tools/profile_round.sh); never measured in this process -> always labelled "from_profile"
        traffic, traffic_src, util = None, None, None
        tj, tsrc, tstale = load_profile_json(f"traffic_{args.workload}.json")
        if tj:
            ks = tj.get("kernels", {})
            if "w_bwd_map" in kernels["bwd"]:       # the reverse pass is several launches: bytes of ONE pass = sum over its launches
                fwd_n = max([v["dispatches"] for k, v in ks.items() if "w_forward" in k or "f_forward" in k] or [0])      # (H = 128 in 1D / 2D: f_forward_coop)
                tot = sum(v["hbm_bytes_per_point"] * v["dispatches"] for k, v in ks.items() if "w_bwd" in k)
                if fwd_n:
                    traffic = tot / fwd_n * n_rows
            else:
                for k, v in ks.items():
                    if "f_backward" in k and (traffic is None or v["hbm_bytes_per_point"] * n_rows > traffic):
                        traffic = v["hbm_bytes_per_point"] * n_rows
            if traffic is not None:
                traffic_src = f"from_profile: {tsrc} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 1/2-fetch correction)" + \
                              (f" -- STALE: recorded on the {tstale} build" if tstale else "")
        uj, usrc, ustale = load_profile_json(f"mfma_util_{args.workload}.json")
        if uj:
            util = dict(uj, source=f"from_profile: {usrc}" + (f" -- STALE: recorded on the {ustale} build" if ustale else ""))
        aj, asrc, astale = load_profile_json(f"accuracy_{args.workload}.json")
        out = {
            "metric": "collocation-point residual evals/sec (full training step: jets fwd + residual + reverse + Adam)",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "timing": {"blocks": len(block_s), "steps_per_block": args.steps, "statistic": "median block (value, ms_per_step)",
                       "ms_per_step_min": min(block_s) / args.steps * 1e3, "ms_per_step_max": max(block_s) / args.steps * 1e3,
                       "ms_per_step_first_block": block_s[0] / args.steps * 1e3, "timed_seconds": float(sum(block_s))},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "layers": layers, "points_per_gpu": n_local,
                       "global_points": n_local * world, "gamma": wl["gamma"], "boundary_points": int(xb.shape[0]),
                       "parallelism": f"dp{world}", "kernel_path": "fused_mfma_f32_16x16x4" if fused else "generic_layerwise_mfma",
                       "exchange": (("engine_rccl_stale1" if args.async_grad else "engine_rccl") if native else "torch_distributed") if use_dist else "none"},
            "per_gpu_points_per_s": value / world,
            "rccl_ranks": comm["world"] if native else (world if use_dist else 0),
            "collectives_per_step": ((comm["collectives"] - coll0) / max(1, steps_from_init)) if native else (2 if use_dist else 0),
            "exchange_crosscheck": xcheck,
            "trajectory": {"steps_from_seeded_init": steps_from_init, "loss": sc["loss"], "mu": sc["mu"],
                           "note": "state after the warm-up and timed steps from a random init -- NOT a converged eigenvalue; the "
                                   "converged accuracy is mu_abs_err (tools/accuracy_nd.py), the correctness of the kernels parity_check"},
            "mu_abs_err": (aj or {}).get("mu_abs_err"), "mu_ref": (aj or {}).get("mu_ref"),
            "mu_abs_err_source": f"from_profile: {asrc} (converged run of tools/accuracy_nd.py vs oracle/gp_ground_state_nd.py)" if aj else None,
            "mu_abs_err_round": (f"{astale} (stale: a converged run of the previous round's build)" if astale else PROFILE_ROUND) if aj else None,
            "roofline": {"bound": "mfma", "kernel": kernels["bwd"],
                         "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP32_MFMA_PEAK_TFLOPS, "frac_gemm_only": ach_gemm / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ((bmat_pt / 2.0 * (Lh - 1) / Lh + 4.0 * chan * layers[-1] + 4.0 * d_in) if "w_bwd_map" not in kernels["bwd"]
                                                          else (4.0 * chan * Hh * (3 * Lh - 5) + 4.0 * chan * layers[-1] + 4.0 * d_in)) * n_rows,
                         "algorithmic_flop_per_point": 2.0 * f_fwd, "gemm_flop_per_point": 2.0 * gemm_fwd,
                         "rows_per_launch": n_rows, "avg_launch_ms": bwd_s * 1e3, "launches": prof["bwd_launches"],
                         "counters": util},
            "roofline_forward": {"bound": "mfma", "kernel": kernels["fwd"], "achieved": ach_f,
                                 "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_f / FP32_MFMA_PEAK_TFLOPS,
                                 "algorithmic_flop_per_point": f_fwd, "avg_launch_ms": fwd_s * 1e3},
            "jet_channels": chan, "survey_8d_step_flop_per_point": 3.0 * survey_fwd,
            "step_flop_per_point": flops_pt, "whole_step_tflops": flops_pt * value / world / 1e12,
            "b_mat_bytes_per_point": bmat_pt, "b_mat_gbps": bmat_pt * value / world / 1e9,
        }
        if not fused:       # layer-wise generic set (MFMA maps for wide layers): no per-kernel events, the whole step is rated
            tf = flops_pt * value / world / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "generic layer-wise set (%s, %s + head/activation kernels), whole step" % (kernels["fwd"], kernels["bwd"]),
                               "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
                               "traffic": None, "algorithmic_flop_per_point": flops_pt, "avg_launch_ms": elapsed / args.steps * 1e3}
            out.pop("roofline_forward", None)
        if world == 1 and fused and not args.no_alt_mode and max(layers[1:-1]) <= 64 and not os.environ.get("GPE_FWD_B6") \
                and not os.environ.get("GPE_BWD_B6"):
            out["split_bf16_mode"] = alt = split_bf16_mode(cfg, flat, x, xb, local_rank, args, n_local)
            # its accounting, over the WHOLE step (no per-kernel events in this pass): the hidden-hidden products of the forward pass and
            # the adjoint products of the reverse pass run as six bf16 products each (executed bf16 MFMA FLOP = 6 x their fp32 FLOP,
            # against the 2.5 PFLOP/s dense bf16 peak); the weight-gradient products stay on the fp32 instruction (157.3 TFLOP/s)
            hh = chan * 2.0 * Hh * Hh * (Lh - 1)                      # fp32 FLOP per point of the hidden-hidden products of ONE pass direction
            alt["accounting"] = {"bf16_mfma_flop_per_point": 6.0 * 2.0 * hh, "fp32_mfma_flop_per_point": hh,
                                 "bf16_mfma_tflops_whole_step": 6.0 * 2.0 * hh * alt["value"] / 1e12, "bf16_peak_tflops": 2500.0,
                                 "bf16_frac_whole_step": 6.0 * 2.0 * hh * alt["value"] / 1e12 / 2500.0,
                                 "fp32_mfma_tflops_whole_step": hh * alt["value"] / 1e12,
                                 "fp32_frac_whole_step": hh * alt["value"] / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                 "pipe_time_frac_whole_step": 6.0 * 2.0 * hh * alt["value"] / 1e12 / 2500.0 + hh * alt["value"] / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                 "note": "fractions of each pipe's peak summed = share of the step the matrix pipes are busy at peak rates"}
        if world == 1 and not args.no_parity_check:
            n_chk = args.parity_points if args.parity_points > 0 else parity_points(layers, n_local, args.parity_budget)
            out["parity_check"] = parity_check(eng, wl, flat, x, dx, xb, min(n_chk, n_local), n_local * world, kernels)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, flat)
            out["cpu_baseline_native"] = cpu_baseline_native(wl, flat)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        eng.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
