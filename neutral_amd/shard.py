"""Particle shards and the end-of-step tally exchange for multi-GPU runs.

The reference has no working distributed path (its MPI blocks are compiled
out, neutral_data.h:10-14; main.c:42-43 hard-wires one rank).  Histories are
independent and keyed by the *global* particle id (omp3/neutral.c:86-89), so
the natural MI355X decomposition is: contiguous id ranges per GPU (the same
split as the OpenMP static partition, omp3/neutral.c:64-74), replicated mesh and
tables, a private per-step tally on each GPU, and ONE all-reduce (sum, f64,
nx*ny elements: 1.28 MB at 400^2) of that tally per timestep -- RCCL over xGMI
when the tensors live in HBM (`backend="nccl"`), gloo for CPU tests.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(nparticles: int, rank: int, world_size: int) -> Tuple[int, int]:
    """(first, count) of the contiguous global-id range owned by `rank`: the
    first `nparticles % world_size` ranks get one extra particle
    (omp3/neutral.c:64-74 does the same over threads)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    per, rem = divmod(int(nparticles), int(world_size))
    first = rank * per + min(rank, rem)
    return first, per + (1 if rank < rem else 0)


class StepTallyExchange:
    """Per-step tally exchange.

    Each rank's kernels add into `step_tally` (zeroed at the start of the
    step); `finish_step()` all-reduces it and accumulates it into `tally`, which
    then holds the same global energy-deposition mesh on every rank -- what a
    single-GPU run accumulates directly.  With world_size == 1 there is no
    collective and kernels may add straight into `tally`.
    """

    def __init__(self, tally, world_size: int, group=None):
        self.tally = tally
        self.world_size = world_size
        self.group = group
        self.step_tally = tally if world_size == 1 else tally.new_zeros(tally.shape)

    def begin_step(self):
        if self.world_size > 1:
            self.step_tally.zero_()
        return self.step_tally

    def finish_step(self):
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.step_tally, op=dist.ReduceOp.SUM, group=self.group)
            self.tally += self.step_tally
        return self.tally
