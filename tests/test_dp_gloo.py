"""world_size = 2 over gloo on the CPU: the data-parallel protocol of the step (dp.py) with the CPU oracle standing in
for the HIP engine (tests may use the oracle as a checker/stand-in; the product never does).  Verifies that the
sharded, all-reduced step equals the single-process full-batch step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gpe_oracle as go
import gpe_pinn
from gpe_pinn.dp import distributed_step, shard_bounds, shard_points


class OracleShardEngine:
    """Same three-phase surface and exchange buffers as gpe_pinn.Engine, computed by oracle/gpe_oracle.py."""

    def __init__(self, pb, flat, x, x_bc, world):
        self.pb, self.flat, self.x, self.x_bc, self.world = pb, flat.astype(np.float32), x.astype(np.float32), x_bc, world
        P = go.param_count(pb.layers)
        self.exchange_sums = torch.zeros(12, dtype=torch.float64)        # the engine's layout (csrc/gpe_common.h: S_NUM .. S_RZ_L, S_COUNT = 12)
        self.exchange_grad = torch.zeros(P + 4, dtype=torch.float32)
        self.opt = go.OptState(lr0=1e-3)
        self.trace = []

    SLOT = dict(num=0, den=1, sym=2, rz_k=7, rz_p=8, rz_i=9, rz_l=10)

    def step_begin(self):
        s = go.loss_and_grad(self.pb, self.flat, self.x, self.x_bc, phase=1)
        self.exchange_sums.zero_()
        for k, i in self.SLOT.items():
            if k in s:
                self.exchange_sums[i] = s[k]

    def step_backward(self):
        tot = {k: float(self.exchange_sums[i]) for k, i in self.SLOT.items()}
        self.res = go.loss_and_grad(self.pb, self.flat, self.x, self.x_bc, shard_sums=tot, phase=2)
        g = self.res["grad_local"] + self.res["grad_bc"] / self.world      # boundary batch is replicated
        self.exchange_grad[:-4] = torch.from_numpy(g.astype(np.float32))
        self.exchange_grad[-4] = self.res["sum_r2"]
        self.exchange_grad[-3:] = 0

    def step_update(self):
        grad = self.exchange_grad[:-4].numpy().astype(np.float64)
        sc = go.assemble(self.pb, self.res, sum_r2_total=float(self.exchange_grad[-4]), n_global=self.pb.n_global)
        self.flat, gn, lr = go.optimizer_step(self.opt, self.flat, grad, sc["loss"])
        sc["grad_norm"] = gn
        self.trace.append(sc)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(N, kind="1d_sym"):
    if kind == "2d_class_loss":          # src/gross_pitaevskii_2D.py:154-242: the lambda branch of the gradient is built from the EXCHANGED energy sums
        return go.Problem(layers=[2, 16, 16, 1], gamma=20.0, kinetic_coeff=1.0, pot_scale=1.0, w_norm=0.0, w_riesz=0.05, riesz_kind=go.RIESZ_SUM,
                          lambda_kind=go.LAMBDA_ENERGY, w_reg_f=1.0, w_reg_lam=1.0, dx=1.0, n_global=N)
    return go.Problem(layers=[1, 16, 16, 1], gamma=3.0, p=3, base_mode=0, dx=12.0 / (N - 1), w_sym=5.0, n_global=N)


def _data(N, kind):
    rng = np.random.default_rng(0)
    pb = _problem(N, kind)
    flat = rng.normal(0, 0.3, go.param_count(pb.layers))
    if kind == "2d_class_loss":
        return pb, flat, rng.uniform(-2, 2, (N, 2)), rng.uniform(-2, 2, (5, 2))
    return pb, flat, np.linspace(-6, 6, N).reshape(-1, 1), np.array([[-6.0], [6.0]])


def _worker(rank, world, port, N, steps, out, kind="1d_sym"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    pb, flat, x, x_bc = _data(N, kind)
    eng = OracleShardEngine(pb, flat, shard_points(x, rank, world), x_bc, world)
    for _ in range(steps):
        distributed_step(eng)
    if rank == 0:
        np.savez(out, flat=eng.flat, loss=[t["loss"] for t in eng.trace], mu=[t["mu"] for t in eng.trace],
                 gn=[t["grad_norm"] for t in eng.trace])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            bounds = [shard_bounds(n, r, w) for r in range(w)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(w - 1))
            sizes = [b[1] - b[0] for b in bounds]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind", ["1d_sym", "2d_class_loss"])
def test_two_rank_step_equals_full_batch(tmp_path, kind):
    N, steps, world = 301, 4, 2                      # odd N: ragged shards
    out = str(tmp_path / "dp.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, steps, out, kind), nprocs=world, join=True)
    got = np.load(out)
    # single-process reference: the same oracle on the full batch
    pb, flat, x, x_bc = _data(N, kind)
    st = go.OptState(lr0=1e-3)
    ref_flat, trace = go.train_steps(pb, st, flat, x, steps, x_bc=x_bc)
    np.testing.assert_allclose(got["loss"], [t["loss"] for t in trace], rtol=2e-5)
    np.testing.assert_allclose(got["mu"], [t["mu"] for t in trace], rtol=2e-5)
    np.testing.assert_allclose(got["gn"], [t["grad_norm"] for t in trace], rtol=2e-4)
    assert np.abs(got["flat"] - ref_flat).max() < 5e-5
