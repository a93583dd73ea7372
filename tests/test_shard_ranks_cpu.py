"""The N > 1 path on the CPU, through the C rank layer: two and three processes, each
advancing the particle shard `comms_shard_range` gives it (the CPU oracle stands in for
the kernels), one tally all-reduce per timestep over the rank layer's links -- the
sequence solve_transport_2d runs on the device -- must reproduce the unsharded run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob
from conftest import ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ranks_reproduce_the_single_rank_run(make_problem, cs, tmp_path, world):
    prob = make_problem("csp", nx=48, nparticles=6001, iterations=3, dt=2.0e-6)
    ref = ob.OracleRun(prob, *cs)
    ref.inject()
    tot = np.zeros(3, dtype=np.uint64)
    for tt in range(1, 4):
        r = ref.step(tt)
        tot += np.array([r.facets, r.collisions, r.nprocessed], dtype=np.uint64)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NEUTRAL_COMM_PORT=str(port),
                   NEUTRAL_COMM_TIMEOUT="60")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests",
                                                                    "shard_worker_cpu.py"),
                                       prob.deck, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, (r, out[-1500:], err[-3000:])
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    spans = [tuple(int(v) for v in r["shard"]) for r in ranks]
    assert spans[0][0] == 0 and spans[-1][0] + spans[-1][1] == 6001
    for (a, c), (b, _) in zip(spans[:-1], spans[1:]):
        assert a + c == b
    for r in ranks:
        assert np.array_equal(r["tally"], ranks[0]["tally"])     # the same bits on every rank
        assert np.array_equal(r["events"], tot)                   # exact event totals
    t = ranks[0]["tally"]
    assert np.linalg.norm(t - ref.tally) / np.linalg.norm(ref.tally) < 1e-12
