"""Data-parallel protocol of the training step: collocation points are sharded in contiguous blocks,
one process per GPU; per step two all-reduces (RCCL when the process group backend is "nccl"):
  1. 8 doubles  [sum u*Hu, sum u^2, sum sym, sum orth_k ...]  -> global Rayleigh quotient mu and norm integral
  2. P+4 floats [flat gradient | sum r^2]                      -> global gradient, pde loss
clip + Adam then run redundantly (bit-identically) on every rank.  The reference is single-device
(refine/harmonic_pinn_simulation.py:12); this module is the build's addition.

`engine` is anything with step_begin/step_backward/step_update and the two exchange tensors -- the HIP Engine in
production; tests drive the same protocol over gloo with a CPU stand-in.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous block partition (grid order keeps the dx quadrature additive).  Blocks differ by at most one point."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_points(x: np.ndarray, rank: int, world: int) -> np.ndarray:
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


def distributed_step(engine, group=None):
    """One synchronous data-parallel step.  Returns nothing; read scalars from the engine afterwards."""
    import torch.distributed as dist
    engine.step_begin()
    dist.all_reduce(engine.exchange_sums, op=dist.ReduceOp.SUM, group=group)
    engine.step_backward()
    dist.all_reduce(engine.exchange_grad, op=dist.ReduceOp.SUM, group=group)
    engine.step_update()
