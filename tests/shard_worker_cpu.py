"""One rank of a sharded run on the CPU (started by tests/test_shard_ranks_cpu.py with
RANK / WORLD_SIZE / MASTER_PORT set): the rank layer of the host library (C, over TCP)
gives the rank its particle shard and sums the per-step tallies; the CPU oracle stands
in for the kernels.  Leaves the global tally and the event totals in <out>/rank<r>.npz."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import oracle_binding as ob  # noqa: E402
from neutral_amd import cs_table, host  # noqa: E402


def main():
    deck, out = sys.argv[1], sys.argv[2]
    L = host.lib()
    L.comms_shard_range.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong),
                                    C.POINTER(C.c_longlong)]
    L.comms_allreduce_f64.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.comms_allreduce_u64.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.comms_start_from_env()
    rank, world = L.comms_rank(), L.comms_nranks()
    prob = host.setup_problem(deck)
    keys, values = cs_table.load()
    first, count = C.c_longlong(), C.c_longlong()
    L.comms_shard_range(prob.nparticles, rank, world, C.byref(first), C.byref(count))
    ob.lib().orc_set_num_threads(2)
    run = ob.OracleRun(prob, keys, values, shard=(first.value, count.value))
    run.inject()
    tally = np.zeros(prob.nx * prob.ny)
    events = np.zeros(3, dtype=np.uint64)
    for tt in range(1, prob.niters + 1):
        run.tally = np.zeros(prob.nx * prob.ny)      # this step's contributions
        r = run.step(tt)
        L.comms_allreduce_f64(run.tally.ctypes.data, run.tally.size, 0)   # COMMS_SUM
        tally += run.tally                            # what solve_transport_2d does on the device
        events += np.array([r.facets, r.collisions, r.nprocessed], dtype=np.uint64)
    L.comms_allreduce_u64(events.ctypes.data, 3, 0)
    np.savez(os.path.join(out, f"rank{rank}.npz"), tally=tally, events=events,
             shard=np.array([first.value, count.value]))
    L.comms_barrier()
    L.comms_stop()


if __name__ == "__main__":
    main()
