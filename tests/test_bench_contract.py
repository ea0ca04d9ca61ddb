"""bench.py keeps the driver's contract: one JSON line with the agreed keys (checked on the GPU with a short run), and its
workload generators are consistent (CPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workloads_are_consistent():
    import bench
    for name, wl in bench.WORKLOADS.items():
        x, dx, xb = bench.make_points(wl, 0, 1)
        d = wl["layers"][0]
        assert x.shape == (int(np.prod(wl["grid"])), d) and xb.shape[1] == d and dx > 0, name
        assert x.dtype == np.float32 and xb.dtype == np.float32
        # rank shards tile the global grid along the first axis
        x0, _, _ = bench.make_points(wl, 0, 2)
        x1, _, _ = bench.make_points(wl, 1, 2)
        assert x0.shape == x.shape and x1.shape == x.shape and x0[:, 0].max() < x1[:, 0].min(), name
        flat = bench.reference_init(wl["layers"])
        n_par = sum(wl["layers"][i] * wl["layers"][i + 1] + wl["layers"][i + 1] for i in range(len(wl["layers"]) - 1))
        assert flat.shape == (n_par,) and np.isfinite(flat).all()


def test_in_run_oracle_check_covers_the_timed_batch_where_the_oracle_fits():
    """bench.parity_points: the whole bound batch for the headline workload (and every workload whose numpy oracle fits the time budget),
    else the largest 4096-multiple slice inside it; stale profile records are labelled."""
    import bench
    for name, full in (("ns_2d_4x64", True), ("cfg1_1d_4x32", True), ("cfg2_1d_4x64", True), ("cfg3_2d_5x128", True), ("cfg5_3d_6x256", False)):
        wl = bench.WORKLOADS[name]
        n = int(np.prod(wl["grid"]))
        k = bench.parity_points(wl["layers"], n, 90.0)
        assert (k == n) == full, (name, k, n)
        assert min(4096, n) <= k <= n and (k == n or k % 4096 == 0)
    rec, path, stale = bench.load_profile_json("no_such_record.json")
    assert rec is None and path is None and stale is None
    rec, path, stale = bench.load_profile_json("traffic_ns_2d_4x64.json")
    assert rec is not None and path.startswith("profiles/") and (stale is None) == path.startswith(f"profiles/{bench.PROFILE_ROUND}/")


def test_driver_history_container():
    from gpe_pinn.surface import _History
    h = _History([{"loss": 1.0, "mu": 2.0, "lr": 1e-3}, {"loss": 0.5, "mu": 2.5, "lr": 1e-3}])
    assert len(h) == 2 and h[1] == {"loss": 0.5, "mu": 2.5, "lr": 1e-3} and h[-1]["mu"] == 2.5
    assert [r["loss"] for r in h] == [1.0, 0.5] and list(h["mu"]) == [2.0, 2.5]
    with pytest.raises(IndexError):
        h[2]
    assert len(_History([])) == 0 and not _History([])
    # ... and from the engine's [n, fields] array (no per-epoch objects: the 201-stage continuation reads 400 000 records)
    import numpy as np
    a = _History(np.array([[1.0, 2.0], [0.5, 2.5], [0.25, 3.0]]), ("loss", "mu"))
    assert len(a) == 3 and a[2] == {"loss": 0.25, "mu": 3.0} and a["loss"][::2].tolist() == [1.0, 0.25] and [r["mu"] for r in a] == [2.0, 2.5, 3.0]


def test_gpus_n_launches_n_ranks():
    """`python bench.py --gpus N` without RANK in the environment starts N worker processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set) before anything touches the GPU; with RANK set (torchrun) it is itself a worker."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["GPE_BENCH_LAUNCH_TEST"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(r["rank"] for r in recs) == [0, 1, 2] and all(r["world"] == 3 and r["local_rank"] == r["rank"] for r in recs)
    assert all(r["master"] == "127.0.0.1" and r["gpus_arg"] == 3 for r in recs)
    env2 = dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=env2)
    recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(recs) == 1 and recs[0]["rank"] == 1 and recs[0]["world"] == 2      # no second launch under torchrun


def test_launcher_ends_the_job_when_a_rank_dies():
    """One rank exits 3 while the others sit in what would be a collective: the launcher notices, terminates the siblings, names the
    rank and exits non-zero within seconds -- it does not wait for the hung ranks; a job in which nobody finishes ends at the time limit."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["GPE_BENCH_LAUNCH_TEST"] = "fail:1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])
    assert "rank 1 exited with code 3" in out.stderr and time.time() - t0 < 60
    env["GPE_BENCH_LAUNCH_TEST"] = "hang"
    env["GPE_BENCH_LAUNCH_TIMEOUT"] = "3"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 124 and "time limit" in out.stderr and time.time() - t0 < 60


def test_scaling_default_is_strong_for_the_sharded_baseline_configs_at_8_gpus():
    """--gpus 8 on cfg3 / cfg4 / cfg5 measures BASELINE's global sizes (strong scaling) unless --scaling says otherwise."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["GPE_BENCH_LAUNCH_TEST"] = "1"
    for wl, gpus, extra, want in (("cfg3_2d_5x128", 8, [], "strong"), ("cfg3_2d_5x128", 2, [], "weak"), ("ns_2d_4x64", 8, [], "weak"),
                                  ("cfg5_3d_6x256", 8, ["--scaling", "weak"], "weak")):
        env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(gpus))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", wl] + extra,
                             capture_output=True, text=True, timeout=120, cwd=ROOT, env=env2)
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert rec["scaling"] == want, (wl, gpus, rec)


def test_strong_scaling_splits_the_baseline_global_grid():
    import bench
    for name in ("ns_2d_4x64", "cfg3_2d_5x128", "cfg5_3d_6x256"):
        wl = bench.WORKLOADS[name]
        tot = int(np.prod(wl["global_grid"]))
        parts = [bench.make_points(wl, r, 8, "strong") for r in range(8)]
        assert sum(p[0].shape[0] for p in parts) == tot
        assert all(abs(p[1] - parts[0][1]) < 1e-12 for p in parts)                           # one quadrature weight
        xs = np.concatenate([p[0][:, 0] for p in parts])
        assert np.all(np.diff(xs[::parts[0][0].shape[0] // wl["global_grid"][0] or 1]) >= -1e-6)   # blocks ordered along the first axis
    g3 = bench.make_points(bench.WORKLOADS["cfg3_2d_5x128"], 0, 8, "strong")[0].shape[0]
    assert g3 == 131072                                                                          # BASELINE configs[2]: 1 048 576 / 8


@pytest.mark.gpu
def test_bench_native_exchange_path_world_1():
    """RANK set (as under torchrun): torch.distributed(nccl) for the rendezvous + the engine's own RCCL communicator for the data path."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577",
               GPE_BENCH_FORCE_XCHECK="1")           # also run the N > 1 cross-check of the two exchange paths (here at world 1)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2_1d_4x64", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["exchange"] == "engine_rccl" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert abs(d["collectives_per_step"] - 2.0) < 1e-9
    x = d["exchange_crosscheck"]
    assert x["ok"] is True and x["grad_rel"] < 1e-5 and x["loss_rel"] < 1e-5, x


@pytest.mark.gpu
def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2_1d_4x64", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 1e7 and abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["points_per_gpu"] - 1.0) < 1e-6
    # the timed block is repeated; the line reports the median block with min / max, and carries in-run correctness evidence
    t = d["timing"]
    assert t["blocks"] >= 5 and t["steps_per_block"] == 3 and t["ms_per_step_min"] <= d["ms_per_step"] <= t["ms_per_step_max"]
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["grad_max_err_over_max_abs"] < 5e-5 and pc["mu_rel_err"] < 2e-5, pc
    # ... taken on the batch AS TIMED, through the kernels that were timed (not on a slice with another launch configuration)
    assert pc["points"] == d["config"]["points_per_gpu"] and pc["same_kernels_as_timed"] is True, pc
    assert "mu_after_timed_steps" not in d and "trajectory" in d
