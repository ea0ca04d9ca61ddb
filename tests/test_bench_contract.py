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
