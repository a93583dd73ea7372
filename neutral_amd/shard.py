"""Which particles a rank owns when the particles are sharded over GPUs.

Histories are independent and keyed by the *global* particle id
(omp3/neutral.c:86-89), so the path shards: contiguous id ranges per rank (the
OpenMP static split of omp3/neutral.c:64-74 over ranks), replicated mesh and tables,
ONE all-reduce of the step's tally per timestep.  The product does all of that in C
inside libneutral_hip.so (host/comms_ranks.c: comms_shard_range, csrc/neutral_comm.hip:
the RCCL exchange); this module is the same partition rule for Python callers
(bench.py, the tests), checked against the C one in tests/test_shard_ranks_cpu.py.
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
