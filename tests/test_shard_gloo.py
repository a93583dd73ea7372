"""N > 1 on CPU with torch.distributed: two gloo ranks, each advancing its particle
shard (the CPU oracle stands in for the kernels; the partition is the product's
neutral_amd.shard.shard_range), one tally all-reduce per timestep -- must reproduce the
unsharded run.  (The product's own exchange is C inside libneutral_hip.so; its CPU
tests are tests/test_comms_ranks.py and tests/test_shard_ranks_cpu.py.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neutral_amd.shard import shard_range


class StepTallyExchange:
    """Test-only: the per-step exchange as torch.distributed would do it.  Each rank's
    "kernels" add into `step_tally` (zeroed at the start of the step); finish_step()
    all-reduces it and accumulates it into `tally`, which then holds the same global
    mesh on every rank."""

    def __init__(self, tally, world_size: int):
        self.tally = tally
        self.world_size = world_size
        self.step_tally = tally if world_size == 1 else tally.new_zeros(tally.shape)

    def begin_step(self):
        if self.world_size > 1:
            self.step_tally.zero_()
        return self.step_tally

    def finish_step(self):
        if self.world_size > 1:
            dist.all_reduce(self.step_tally, op=dist.ReduceOp.SUM)
            self.tally += self.step_tally
        return self.tally


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1000, 10**8 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (a, c), (b, _) in zip(spans[:-1], spans[1:]):
                assert a + c == b
            assert spans[-1][0] + spans[-1][1] == n
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_shard_range_is_the_c_rank_layers_partition():
    """neutral_amd.shard.shard_range == comms_shard_range (host/comms_ranks.c), which is
    what inject_particles cuts a rank's share with."""
    import ctypes as C
    from neutral_amd import host
    L = host.lib()
    L.comms_shard_range.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong),
                                    C.POINTER(C.c_longlong)]
    for n in (0, 1, 7, 200001, 10**8 + 3):
        for world in (1, 2, 3, 8):
            for r in range(world):
                first, count = C.c_longlong(), C.c_longlong()
                L.comms_shard_range(n, r, world, C.byref(first), C.byref(count))
                assert (first.value, count.value) == shard_range(n, r, world)


def _worker(rank, world, port, deck_path, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_binding as ob
    from neutral_amd import cs_table, host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ob.lib().orc_set_num_threads(2)
    prob = host.setup_problem(deck_path)
    keys, values = cs_table.load()
    first, count = shard_range(prob.nparticles, rank, world)
    run = ob.OracleRun(prob, keys, values, shard=(first, count))
    run.inject()
    global_tally = torch.zeros(prob.nx * prob.ny, dtype=torch.float64)
    exchange = StepTallyExchange(global_tally, world)
    events = torch.zeros(3, dtype=torch.float64)
    for tt in range(1, prob.niters + 1):
        step_buf = exchange.begin_step()
        run.tally = step_buf.numpy()          # the "kernel" adds into the step buffer
        r = run.step(tt)
        exchange.finish_step()
        events += torch.tensor([r.facets, r.collisions, r.nprocessed], dtype=torch.float64)
    dist.all_reduce(events)
    np.save(os.path.join(out_dir, f"tally_{rank}.npy"), global_tally.numpy())
    np.save(os.path.join(out_dir, f"events_{rank}.npy"), events.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_run_reproduces_single_rank(make_problem, cs, tmp_path):
    import oracle_binding as ob
    prob = make_problem("csp", nx=48, nparticles=6001, iterations=3, dt=2.0e-6)
    ref = ob.OracleRun(prob, *cs)
    ref.inject()
    tot = np.zeros(3)
    for tt in range(1, 4):
        r = ref.step(tt)
        tot += (r.facets, r.collisions, r.nprocessed)
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, prob.deck, str(tmp_path)), nprocs=2, join=True)
    t0 = np.load(tmp_path / "tally_0.npy")
    t1 = np.load(tmp_path / "tally_1.npy")
    assert np.array_equal(t0, t1)                      # every rank holds the global tally
    assert np.linalg.norm(t0 - ref.tally) / np.linalg.norm(ref.tally) < 1e-12
    assert np.array_equal(np.load(tmp_path / "events_0.npy"), tot)   # exact event totals
