"""The headline configuration at its FULL size against the CPU oracle, and the at-scale check of
the collision stage's time slicing + work stealing -- part of the default GPU suite (`-m gpu`)
wherever the host can carry them: >= 16 usable cores and >= 48 GB of free memory (the pair is
minutes of host CPU and ~25 GB of host memory); elsewhere they skip and say why.
`NEUTRAL_FULL_SCALE=0` opts out, `NEUTRAL_FULL_SCALE=1` forces them on any host.

  python -m pytest tests/test_full_scale.py -m gpu -x -q -s

The reference validates at full deck size (omp3/neutral.c:520-557, problems/neutral.tests:1-3);
the rest of the suite compares with the oracle at 1e6 particles and checks size-independent
properties at 1e8 (tests/test_hip_parity.py).  Logs: profiles/r04/full_scale.log, profiles/r05/."""
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
from conftest import gpu_available

MIN_CORES = 16  # (what a one-GPU box of the pool gives a job: cpu.max = 16 of its 256 threads)
MIN_FREE_GB = 48.0


def usable_cores() -> int:
    """Cores this process may run on: its affinity mask, cut down by a cgroup CPU quota if one
    is set (a container's `cpu.max`)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                words = f.read().split()
            if path.endswith("cpu.max"):
                if words and words[0] != "max":
                    n = min(n, max(1, int(int(words[0]) / int(words[1]))))
            else:
                quota = int(words[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, quota // int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return n


def free_memory_gb() -> float:
    """MemAvailable, cut down by what a cgroup memory limit leaves."""
    free = 0.0
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    free = int(line.split()[1]) / 1048576.0
    except OSError:
        return 0.0
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            limit = f.read().strip()
        if limit != "max":
            with open("/sys/fs/cgroup/memory.current") as f:
                used = int(f.read().strip())
            free = min(free, (int(limit) - used) / 2.0**30)
    except (OSError, ValueError):
        pass
    return free


def _why_not(min_cores, min_free_gb):
    """None when a full-scale test runs here, else the reason it does not."""
    force = os.environ.get("NEUTRAL_FULL_SCALE")
    if force == "0":
        return "NEUTRAL_FULL_SCALE=0"
    if force == "1":
        return None
    cores, free = usable_cores(), free_memory_gb()
    if cores < min_cores or free < min_free_gb:
        return (f"host too small: {cores} usable cores (need {min_cores}), {free:.0f} GB free "
                f"(need {min_free_gb:.0f}); NEUTRAL_FULL_SCALE=1 forces it")
    return None


# the oracle at 1e8 particles: minutes on 32+ cores, ~25 GB; the bitwise pair at 4e7: two copies
# of the particle arrays on the host (6 GB), no oracle
_WHY_NOT_ORACLE = _why_not(MIN_CORES, MIN_FREE_GB) if gpu_available() else "needs a GPU"
_WHY_NOT_BITWISE = _why_not(1, 16.0) if gpu_available() else "needs a GPU"

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]


_CAPSYS = None


def _progress(msg):
    """A line the runner sees while pytest holds the output back (minutes pass between dots
    here): written with the capture switched off (pytest captures at the file-descriptor level,
    so sys.__stderr__ is held back like everything else)."""
    if _CAPSYS is not None:
        with _CAPSYS.disabled():
            print(f"[full scale] {msg}", flush=True)
    else:
        print(msg, flush=True)


@pytest.fixture(autouse=True)
def _uncaptured(capsys):
    global _CAPSYS
    _CAPSYS = capsys
    yield
    _CAPSYS = None


TALLY_L2_TOL = 1e-9   # bar: 1e-6 (BASELINE.json north_star); floating state: 1e-9
STATE_TOL = 1e-9


@pytest.fixture()
def iface():
    from neutral_amd import interface
    interface.set_quiet(True)
    interface.set_lazy_export(False)
    return interface


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.mark.skipif(_WHY_NOT_ORACLE is not None, reason=str(_WHY_NOT_ORACLE))
def test_headline_size_against_the_oracle(iface, make_problem, cs):
    """csp 400^2 with 1e8 particles (the configuration BASELINE.json's metric is quoted on), four
    timesteps -- the first two are flight (a free flight drawn in the vacuum outlasts its step:
    omp3/neutral.c:127-131; only a history that BEGINS a step inside the dense block collides),
    the next two put the collision stage's time-sliced rings, weighted shares and stealing to
    work (7e9 collisions) -- default pipeline, default write-back, against the oracle on
    the host's cores: per-step (nprocessed, facets, collisions, census) exact, cellx / celly /
    dead exact for every particle, floating state 1e-9 (positions absolute: the mesh is one
    unit wide and 1e8 samples include x = 1e-7), per-cell tally L2 <= 1e-9, same zero pattern."""
    n, steps = 100000000, 4
    prob = make_problem("csp", nx=400, nparticles=n, iterations=steps)
    threads = min(64, usable_cores())
    _progress(f"csp 400^2 / {n} x {steps} steps; oracle on {threads} threads")
    ob.lib().orc_set_num_threads(threads)
    ref = ob.OracleRun(prob, *cs)
    ref.inject()
    sim = iface.Simulation(prob, *cs, variant=2)
    sim.inject()
    steals = 0
    sum_collisions = 0
    for tt in range(1, steps + 1):
        g = sim.step(tt)
        c = ref.step(tt)
        _progress(f"step {tt}: HIP {g.nprocessed} {g.facets} {g.collisions} {g.census} | oracle "
                  f"{c.nprocessed} {c.facets} {c.collisions} {c.census} | steals {g.stats.steals} "
                  f"weighted waves {g.stats.weighted_waves} "
                  f"requeued {g.stats.requeued} passes {g.stats.stream_passes}")
        assert (g.nprocessed, g.facets, g.collisions, g.census) == \
            (c.nprocessed, c.facets, c.collisions, c.census)
        assert g.stats.aborted == 0 and g.stats.steals_refused == 0
        steals += g.stats.steals
        sum_collisions += g.collisions
    tg, tc = sim.tally_host(), ref.tally
    l2 = float(np.linalg.norm(tg - tc) / np.linalg.norm(tc))
    _progress(f"per-cell tally L2 {l2:.3e}; global sum rel "
              f"{abs(tg.sum() - tc.sum()) / abs(tc.sum()):.3e}; steals {steals}")
    assert l2 < TALLY_L2_TOL
    assert np.array_equal(tg == 0.0, tc == 0.0)
    assert sum_collisions > 1000000000 and steals > 0, "the run no longer reaches the dense block"
    gp, cp = sim.particle_arrays(), ref.particles.as_dict()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(gp[f], cp[f]), f
    for f in ("energy", "weight", "dt_to_census"):
        assert _rel(gp[f], cp[f]) < STATE_TOL, f
    for f in ("omega_x", "omega_y", "x", "y"):
        assert float(np.max(np.abs(gp[f] - cp[f]))) < STATE_TOL, f
    _progress(f"{n} particles: cells and death flags equal, floating state within {STATE_TOL}")
    sim.close()


@pytest.mark.skipif(_WHY_NOT_BITWISE is not None, reason=str(_WHY_NOT_BITWISE))
def test_time_slicing_and_stealing_at_scale_are_bitwise(iface, make_problem, cs):
    """csp 400^2, 4e7 particles, 4 timesteps: the tiled pipeline with its defaults (rings of
    hundreds of histories per wave, taken from by CU-mates thousands of times) against the
    over-particle kernel -- same event counts, EVERY particle field bit for bit, tallies equal up
    to summation order.  (tools/micro/compare_variants.py is the same check as a script.)"""
    n, steps = 40000000, 4
    prob = make_problem("csp", nx=400, nparticles=n, iterations=steps)

    def run(variant):
        sim = iface.Simulation(prob, *cs, variant=variant)
        sim.inject()
        ev, steals, refused = [], 0, 0
        for tt in range(1, steps + 1):
            r = sim.step(tt)
            ev.append((r.nprocessed, r.facets, r.collisions, r.census))
            _progress(f"variant {variant} step {tt}: {ev[-1]}")
            steals += r.stats.steals
            refused += r.stats.steals_refused
            assert r.stats.aborted == 0
        out = (ev, sim.particle_arrays(), sim.tally_host(), steals, refused)
        sim.close()
        return out

    ev0, a0, t0, _, _ = run(0)
    ev2, a2, t2, steals, refused = run(2)
    _progress(f"{n} particles x {steps} steps: steals {steals}, refused {refused}")
    assert ev0 == ev2
    for f in a0:
        assert np.array_equal(a0[f], a2[f]), f
    assert float(np.linalg.norm(t0 - t2) / np.linalg.norm(t0)) < 1e-12
    assert steals >= 3000   # (thousands: how many exactly follows the waves' timing -- 8 800 to 15 000 across builds)
    assert refused == 0
