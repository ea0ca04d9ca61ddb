#!/usr/bin/env python3
"""bench.py -- throughput of the GPE eigenvalue-residual training step on MI355X.

A "step" is one full pass of the hot path over the collocation batch: jet forward (psi, grad psi, diag Hessian),
Rayleigh quotient mu, residual, boundary + normalisation penalties, reverse pass, grad-norm clip, Adam, scheduler.

Workload (config.workload = "ns_2d_4x64"): BASELINE.json's north-star configuration -- the one its metric target is
quoted on -- 2D isotropic harmonic trap, g = 500, MLP [2,64,64,64,64,1] (4 hidden x 64), 1 048 576 collocation points
per GPU (1024 x 1024 uniform grid per rank), 512 boundary points; synthetic seeded weights (reference init a13).
With --gpus N every rank owns its own 1 048 576-point shard (weak scaling); the two per-step exchanges
(8 doubles, then P+4 floats) are RCCL all-reduces through torch.distributed.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      -- dominant kernel (fused jet reverse pass) algorithmic FLOP / HIP-event time vs the fp32 MFMA peak
  cpu_baseline  -- the reference's op sequence (torch-autograd restatement, oracle/torch_ref.py) timed on this host
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"

WORKLOADS = {
    # name: (layers, dim, points per GPU (grid), gamma, domain half-width)
    "ns_2d_4x64": dict(layers=[2, 64, 64, 64, 64, 1], grid=(1024, 1024), gamma=500.0, half=8.0),
    "cfg2_1d_4x64": dict(layers=[1, 64, 64, 64, 64, 1], grid=(65536,), gamma=100.0, half=10.0),
    "cfg3_2d_5x128": dict(layers=[2, 128, 128, 128, 128, 128, 1], grid=(512, 256), gamma=500.0, half=8.0),
    # BASELINE configs[3]: rotating trap, complex psi (n_out = 2), 6x128; 2 097 152 points over 8 GPUs = 262 144 per GPU
    "cfg4_2d_6x128_rot": dict(layers=[2, 128, 128, 128, 128, 128, 128, 2], grid=(512, 512), gamma=500.0, half=8.0,
                              complex_psi=True, omega_rot=0.8),
    # BASELINE configs[4] shape: 3D anisotropic trap, 6x256 (generic layer-materialised kernel set; no fused path for H = 256 yet).
    # 4 194 304 points over 8 GPUs = 524 288 per GPU; the grid here is a quarter of that to keep the default run short.
    "cfg5_3d_6x256": dict(layers=[3, 256, 256, 256, 256, 256, 256, 1], grid=(64, 64, 32), gamma=1000.0, half=6.0, generic_ok=True),
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


def make_points(wl, rank, world):
    half = wl["half"]
    if len(wl["grid"]) == 1:
        n = wl["grid"][0]
        xs = np.linspace(-half, half, n * world, dtype=np.float64)
        x = xs[rank * n:(rank + 1) * n].reshape(-1, 1)
        dx = 2 * half / (n * world - 1)
        xb = np.array([[-half], [half]])
    elif len(wl["grid"]) == 3:
        nx, ny, nz = wl["grid"]
        xs = np.linspace(-half, half, nx * world, dtype=np.float64)[rank * nx:(rank + 1) * nx]
        ys = np.linspace(-half, half, ny, dtype=np.float64)
        zs = np.linspace(-half, half, nz, dtype=np.float64)
        X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        dx = (2 * half / (nx * world - 1)) * (2 * half / (ny - 1)) * (2 * half / (nz - 1))
        t = np.linspace(-half, half, 23, endpoint=False)
        A, B = np.meshgrid(t, t, indexing="ij")
        xb = np.stack([A.ravel(), B.ravel(), np.full(A.size, half)], axis=1)          # one face of the box, 529 points
    else:
        nx, ny = wl["grid"]
        xs = np.linspace(-half, half, nx * world, dtype=np.float64)[rank * nx:(rank + 1) * nx]
        ys = np.linspace(-half, half, ny, dtype=np.float64)
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel()], axis=1)
        dx = (2 * half / (nx * world - 1)) * (2 * half / (ny - 1))
        t = np.linspace(-half, half, 128, endpoint=False)
        xb = np.concatenate([np.stack([t, np.full_like(t, -half)], 1), np.stack([np.full_like(t, half), t], 1),
                             np.stack([-t, np.full_like(t, half)], 1), np.stack([np.full_like(t, -half), -t], 1)])
    return x.astype(np.float32), float(dx), xb.astype(np.float32)


def cpu_baseline(wl, flat, budget_s=12.0, n_sample=8192):
    """The reference's own op sequence on the host cores (bounded sample of the same workload)."""
    from oracle import gpe_oracle as go
    from oracle import torch_ref as tr
    x, dx, xb = make_points(wl, 0, 1)
    idx = np.linspace(0, x.shape[0] - 1, n_sample).astype(np.int64)
    xs = x[idx]
    pb = go.Problem(layers=wl["layers"], gamma=wl["gamma"], p=3, kinetic_coeff=0.5, pot_scale=0.5, dx=dx,
                    w_bc=10.0, w_norm=20.0, complex_psi=bool(wl.get("complex_psi", False)),
                    omega_rot=float(wl.get("omega_rot", 0.0)))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ns_2d_4x64", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun always take the RCCL path, even at world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import gpe_pinn
    wl = WORKLOADS[args.workload]
    layers = wl["layers"]
    flat = reference_init(layers, seed=0)
    x, dx, xb = make_points(wl, rank, world)
    n_local = x.shape[0]
    cfg = gpe_pinn.GPEConfig(layers=layers, gamma=wl["gamma"], p=3, kinetic_coeff=0.5, pot_scale=0.5, dx=dx,
                             w_bc=10.0, w_norm=20.0, lr=1e-3, n_global=n_local * world, world_size=world,
                             complex_psi=bool(wl.get("complex_psi", False)), omega_rot=float(wl.get("omega_rot", 0.0)))
    eng = gpe_pinn.Engine(cfg, device=local_rank)
    fused = eng.active_path == gpe_pinn.PATH_FUSED
    if not fused and not wl.get("generic_ok", False):
        raise SystemExit("bench: fused MFMA path not active")
    eng.set_params(flat)
    eng.bind_points(torch.as_tensor(x, device=f"cuda:{local_rank}"))      # inputs resident in HBM before timing
    eng.bind_boundary(torch.as_tensor(xb, device=f"cuda:{local_rank}"))

    def run_steps(k):
        if not use_dist:
            eng.run(k)
        else:
            for _ in range(k):
                eng.step_distributed()

    run_steps(args.warmup)
    eng.synchronize()
    eng.profile_enable(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = eng.profile_read()
    eng.profile_enable(False)
    sc = eng.read_scalars()

    if rank == 0:
        flops_pt, bmat_pt = eng.step_cost()
        f_fwd = flops_pt / 3.0
        # channels the training batches carry: value, d first derivatives, ONE Laplacian channel (SURVEY 8(d) counts 1 + 2d;
        # its per-point figure is reported next to the executed one -- the roofline fraction uses only executed flops)
        d_in, Hh, Lh = layers[0], layers[1], len(layers) - 2
        chan = d_in + 2
        survey_fwd = 2.0 * d_in * Hh + (1 + 2 * d_in) * (2.0 * Hh * Hh * (Lh - 1) + 2.0 * Hh * layers[-1]) + Hh * Lh * (3 + 5 * d_in)
        pts = n_local * world * args.steps
        value = pts / elapsed
        bwd_s = prof["bwd_ms"] / max(1, prof["bwd_launches"]) * 1e-3
        fwd_s = prof["fwd_ms"] / max(1, prof["fwd_launches"]) * 1e-3
        ach = 2.0 * f_fwd * n_local / bwd_s / 1e12 if bwd_s > 0 else 0.0
        ach_f = f_fwd * n_local / fwd_s / 1e12 if fwd_s > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC run of this same workload
        # (profiles/r01/traffic_ns_v7.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 1/2-fetch correction applied)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01", "traffic_ns_v7.json")
        if args.workload == "ns_2d_4x64" and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                for k, v in tj["kernels"].items():          # the collocation batch's launch is the largest f_backward* entry
                    if "f_backward" in k and (traffic is None or v["hbm_bytes_per_point"] * n_local > traffic):
                        traffic = v["hbm_bytes_per_point"] * n_local
                        traffic_src = "profiles/r01/traffic_ns_v7.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
            except Exception:
                pass
        out = {
            "metric": "collocation-point residual evals/sec (full training step: jets fwd + residual + reverse + Adam)",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "layers": layers, "points_per_gpu": n_local,
                       "global_points": n_local * world, "gamma": wl["gamma"], "boundary_points": int(xb.shape[0]),
                       "parallelism": f"dp{world}", "kernel_path": "fused_mfma_f32_16x16x4" if fused else "generic_valu_layerwise"},
            "per_gpu_points_per_s": value / world,
            "final_loss": sc["loss"], "final_mu": sc["mu"],
            "roofline": {"bound": "mfma", "kernel": "%s<%d,%d,...> (fused jet reverse pass, %d channels)" % ("f_backward_coop" if ((layers[1] <= 64 and len(layers) - 3 <= 3) or (layers[1] == 128 and len(layers) - 3 <= 5 and layers[0] <= 2)) else "f_backward", layers[1], chan, chan),
                         "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": (bmat_pt / 2.0 * (len(layers) - 3) / (len(layers) - 2) + 4.0 * chan + 4.0 * layers[0] + 8.0) * n_local,
                         "algorithmic_flop_per_point": 2.0 * f_fwd, "avg_launch_ms": bwd_s * 1e3,
                         "launches": prof["bwd_launches"]},
            "roofline_forward": {"bound": "mfma", "kernel": "f_forward<%d,%d,1>" % (layers[1], chan), "achieved": ach_f,
                                 "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_f / FP32_MFMA_PEAK_TFLOPS,
                                 "algorithmic_flop_per_point": f_fwd, "avg_launch_ms": fwd_s * 1e3},
            "jet_channels": chan, "survey_8d_step_flop_per_point": 3.0 * survey_fwd,
            "step_flop_per_point": flops_pt, "whole_step_tflops": flops_pt * value / world / 1e12,
            "b_mat_bytes_per_point": bmat_pt, "b_mat_gbps": bmat_pt * value / world / 1e9,
        }
        if not fused:       # layer-wise generic set (MFMA maps for wide layers): no per-kernel events, the whole step is rated
            tf = flops_pt * value / world / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "generic layer-wise set (g_*_mfma maps + VALU head/activation kernels), whole step",
                               "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
                               "traffic": None, "algorithmic_flop_per_point": flops_pt, "avg_launch_ms": elapsed / args.steps * 1e3}
            out.pop("roofline_forward", None)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, flat)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
