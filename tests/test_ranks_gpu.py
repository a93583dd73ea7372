"""Several ranks through the C path on a real GPU (`-m gpu`): `neutral.hip --gpus N`
forks one process per rank before anything touches the GPU; the library shards the
particles, every rank steps its shard, and solve_transport_2d ends each timestep
with the all-reduce of the tally.  The test box has ONE GPU, so the ranks share it
(NEUTRAL_HIP_SHARE_DEVICE) and the exchange is staged through the host; RCCL itself
is exercised with a one-rank communicator (load, init, all-reduce on the device).
Every multi-rank run is compared first with the CPU oracle on the same deck (one rank,
whole mesh: per-step event counts exact, the global tally to 1e-10) and then with the
one-rank HIP run."""
import os
import re
import subprocess

import pytest

import oracle_binding as ob
from conftest import ROOT, gpu_available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]

OWN_DRIVER = os.path.join(ROOT, "neutral_amd", "host", "neutral.hip")


def test_rccl_loads_and_reduces_on_this_device():
    from neutral_amd import interface as iface
    iface.set_device(0)
    assert iface.library().neutral_hip_comm_selftest(1 << 16) == 0


def _run_driver(run_dir, rel, extra, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    out = subprocess.run([OWN_DRIVER, rel] + extra, cwd=run_dir, capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout, out.stderr


def _numbers(stdout):
    facets = [int(x) for x in re.findall(r"^Facets\s+(\d+)", stdout, flags=re.M)]
    colls = [int(x) for x in re.findall(r"^Collisions\s+(\d+)", stdout, flags=re.M)]
    parts = [int(x) for x in re.findall(r"^Particles\s+(\d+)", stdout, flags=re.M)]
    tally = float(re.search(r"Final global_energy_tally (\S+)", stdout).group(1))
    return facets, colls, parts, tally


_ORACLE_CACHE = {}


def _oracle_numbers(tmp_path, cs, **overrides):
    """(facets, collisions, particles) per step and the global tally of the CPU oracle
    (oracle/neutral_oracle.c, restating omp3/neutral.c:19-206) on the csp deck."""
    from neutral_amd import decks, host
    key = tuple(sorted(overrides.items()))
    if key not in _ORACLE_CACHE:
        path = decks.write_deck("csp", str(tmp_path / "oracle_csp.params"), **overrides)
        prob = host.setup_problem(path, 1.0, 1.0)
        ref = ob.OracleRun(prob, *cs)
        ref.inject()
        f, c, p = [], [], []
        for tt in range(1, overrides["iterations"] + 1):
            r = ref.step(tt)
            f.append(r.facets)
            c.append(r.collisions)
            p.append(r.nprocessed)
        _ORACLE_CACHE[key] = (f, c, p, ref.tally_sum())
    return _ORACLE_CACHE[key]


CSP_128 = dict(nx=128, ny=128, nparticles=200001, iterations=4, dt=1.0e-6)


@pytest.mark.skipif(not os.path.exists(OWN_DRIVER), reason="neutral.hip not built")
@pytest.mark.parametrize("nranks,comm", [(2, "host"), (3, "host"), (2, "rccl")])
def test_forked_ranks_reproduce_the_one_rank_run(tmp_path, cs, nranks, comm):
    """csp at 128^2 with 200 001 particles (not divisible by the rank count), 4 steps:
    every step's global event counts are exact, the tally agrees to summation order.
    comm = rccl on a shared device: ncclCommInitRank refuses the duplicate GPU, every
    rank notices, and the run falls back to the host route by itself."""
    from neutral_amd import cs_table, decks
    run = tmp_path / "arch" / "neutral"
    (run / "problems").mkdir(parents=True)
    (tmp_path / "arch" / "arch.params").write_text("width 1.0\nheight 1.0\nsim_end 100.0\n")
    cs_table.write_files(str(run))
    rel = os.path.join("problems", "csp.params")
    decks.write_deck("csp", str(run / rel))
    sets = []
    for kv in ("nx=128", "ny=128", "nparticles=200001", "iterations=4", "dt=1.0e-6"):
        sets += ["--set", kv]
    one, _ = _run_driver(str(run), rel, sets)
    env = {"NEUTRAL_HIP_SHARE_DEVICE": "1", "NEUTRAL_COMM_TIMEOUT": "20"}
    if comm == "host":
        env["NEUTRAL_HIP_COMM"] = "host"
    many, err = _run_driver(str(run), rel, sets + ["--gpus", str(nranks)], env)
    f1, c1, p1, t1 = _numbers(one)
    fn, cn, pn, tn = _numbers(many)
    fo, co, po, to = _oracle_numbers(tmp_path, cs, **CSP_128)
    assert (fo, co, po) == (fn, cn, pn), many[-1500:]      # the oracle first
    assert abs(tn - to) <= 1e-10 * abs(to)
    assert (f1, c1, p1) == (fn, cn, pn), (one[-1500:], many[-1500:])
    assert len(f1) == 4
    assert abs(tn - t1) <= 1e-12 * abs(t1)
    assert many.count("Iteration") == 4          # one rank speaks
    assert f"{nranks} ranks, tally exchange over the host" in err


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(not os.path.exists(OWN_DRIVER), reason="neutral.hip not built")
@pytest.mark.skipif(_visible_gpus() < 2, reason="RCCL between GPUs needs two of them (the test box has one)")
def test_two_gpus_exchange_over_rccl(tmp_path, cs):
    """Two ranks on two DIFFERENT GPUs: the tally and the step words are all-reduced by RCCL
    on the library's own stream (no host transport, no host collective); per-step event
    counts equal the CPU oracle's, the global tally to 1e-10."""
    from neutral_amd import cs_table, decks
    run = tmp_path / "arch" / "neutral"
    (run / "problems").mkdir(parents=True)
    (tmp_path / "arch" / "arch.params").write_text("width 1.0\nheight 1.0\nsim_end 100.0\n")
    cs_table.write_files(str(run))
    rel = os.path.join("problems", "csp.params")
    decks.write_deck("csp", str(run / rel))
    sets = []
    for kv in ("nx=128", "ny=128", "nparticles=200001", "iterations=4", "dt=1.0e-6"):
        sets += ["--set", kv]
    many, err = _run_driver(str(run), rel, sets + ["--gpus", "2"], {"NEUTRAL_COMM_TIMEOUT": "120"})
    fn, cn, pn, tn = _numbers(many)
    fo, co, po, to = _oracle_numbers(tmp_path, cs, **CSP_128)
    assert (fo, co, po) == (fn, cn, pn), many[-1500:]
    assert abs(tn - to) <= 1e-10 * abs(to)
    assert "2 ranks, tally exchange over RCCL" in err


@pytest.mark.skipif(not os.path.exists(OWN_DRIVER), reason="neutral.hip not built")
@pytest.mark.skipif(_visible_gpus() < 2, reason="ncclSend/ncclRecv between GPUs needs two of them (the test box has one)")
def test_two_gpus_exchange_particles_over_rccl(tmp_path, cs):
    """`neutral.hip --gpus 2 --decompose 2x1` on two DIFFERENT GPUs: the emigrants of every round
    travel by ncclSend / ncclRecv in one group (neutral_comm.hip: comm_exchange_bytes), not over the
    host links -- so that this branch first runs under a comparison with the CPU oracle and with
    the one-rank run, not under the bench.  Per-step event counts exact, tally 1e-10 / 1e-12.
    (Turns itself on wherever two GPUs are visible; the pool's test box has one.)"""
    from neutral_amd import cs_table, decks
    run = tmp_path / "arch" / "neutral"
    (run / "problems").mkdir(parents=True)
    (tmp_path / "arch" / "arch.params").write_text("width 1.0\nheight 1.0\nsim_end 100.0\n")
    cs_table.write_files(str(run))
    rel = os.path.join("problems", "csp.params")
    decks.write_deck("csp", str(run / rel))
    sets = []
    for kv in ("nx=128", "ny=128", "nparticles=200001", "iterations=4", "dt=1.0e-6"):
        sets += ["--set", kv]
    one, _ = _run_driver(str(run), rel, sets)
    many, err = _run_driver(str(run), rel, sets + ["--gpus", "2", "--decompose", "2x1"],
                            {"NEUTRAL_COMM_TIMEOUT": "120"})
    f1, c1, p1, t1 = _numbers(one)
    fn, cn, pn, tn = _numbers(many)
    fo, co, po, to = _oracle_numbers(tmp_path, cs, **CSP_128)
    assert (fo, co, po) == (fn, cn, pn), many[-1500:]      # the oracle first
    assert abs(tn - to) <= 1e-10 * abs(to)
    assert (f1, c1, p1) == (fn, cn, pn), (one[-1500:], many[-1500:])
    assert abs(tn - t1) <= 1e-12 * abs(t1)
    assert "over RCCL" in err, err[-1500:]


@pytest.mark.skipif(not os.path.exists(OWN_DRIVER), reason="neutral.hip not built")
@pytest.mark.parametrize("nranks,grid", [(2, "2x1"), (4, "2x2")])
def test_forked_ranks_with_a_decomposed_mesh(tmp_path, cs, nranks, grid):
    """`neutral.hip --gpus N --decompose PXxPY`: every rank holds a block of the mesh and
    the particles inside it; histories cross between the blocks within the step.  Event
    counts as in the one-rank run, exactly; the tally (summed over the blocks by
    validate) to summation order."""
    from neutral_amd import cs_table, decks
    run = tmp_path / "arch" / "neutral"
    (run / "problems").mkdir(parents=True)
    (tmp_path / "arch" / "arch.params").write_text("width 1.0\nheight 1.0\nsim_end 100.0\n")
    cs_table.write_files(str(run))
    rel = os.path.join("problems", "csp.params")
    decks.write_deck("csp", str(run / rel))
    sets = []
    for kv in ("nx=128", "ny=128", "nparticles=200001", "iterations=4", "dt=1.0e-6"):
        sets += ["--set", kv]
    one, _ = _run_driver(str(run), rel, sets)
    env = {"NEUTRAL_HIP_SHARE_DEVICE": "1", "NEUTRAL_COMM_TIMEOUT": "60",
           "NEUTRAL_HIP_COMM": "host"}
    many, err = _run_driver(str(run), rel, sets + ["--gpus", str(nranks), "--decompose", grid],
                            env)
    f1, c1, p1, t1 = _numbers(one)
    fn, cn, pn, tn = _numbers(many)
    fo, co, po, to = _oracle_numbers(tmp_path, cs, **CSP_128)
    assert (fo, co, po) == (fn, cn, pn), many[-1500:]      # the oracle first
    assert abs(tn - to) <= 1e-10 * abs(to)
    assert (f1, c1, p1) == (fn, cn, pn), (one[-1500:], many[-1500:])
    assert abs(tn - t1) <= 1e-12 * abs(t1)
