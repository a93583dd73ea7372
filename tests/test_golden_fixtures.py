"""Committed vectors of the pinned oracle (tests/golden/oracle_<deck>.npz, made by
tests/golden/make_oracle_fixtures.py): the oracle must still reproduce them bit
for bit (CPU), and the HIP path must match them without the oracle at hand (GPU)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from conftest import ROOT, gpu_available

DECKS = ["scatter", "stream", "csp", "split"]


def _load(deck):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"oracle_{deck}.npz"))
    nx, n, its = (int(v) for v in z["config"])
    return z, nx, n, its, float(z["dt"])


@pytest.mark.parametrize("deck", DECKS)
def test_oracle_reproduces_its_committed_vectors(make_problem, cs, deck):
    z, nx, n, its, dt = _load(deck)
    prob = make_problem(deck, nx=nx, nparticles=n, iterations=its, dt=dt)
    ob.lib().orc_set_num_threads(1)
    try:
        run = ob.OracleRun(prob, *cs)
        run.inject()
        events = []
        for tt in range(1, its + 1):
            r = run.step(tt)
            events.append((r.nprocessed, r.facets, r.collisions, r.census))
    finally:
        ob.lib().orc_set_num_threads(os.cpu_count() or 1)
    assert np.array_equal(np.array(events), z["events"])
    for f in ob.F64_FIELDS + ob.I32_FIELDS:
        assert np.array_equal(getattr(run.particles, f), z[f]), f
    assert np.array_equal(run.tally, z["tally"])


@pytest.mark.gpu
@pytest.mark.skipif(not gpu_available(), reason="needs a GPU")
@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("deck", DECKS)
def test_hip_path_matches_committed_vectors(make_problem, cs, deck, variant):
    from neutral_amd import interface as iface
    iface.set_quiet(True)
    z, nx, n, its, dt = _load(deck)
    prob = make_problem(deck, nx=nx, nparticles=n, iterations=its, dt=dt)
    sim = iface.Simulation(prob, *cs, variant=variant)
    sim.inject()
    events = []
    for tt in range(1, its + 1):
        r = sim.step(tt)
        events.append((r.nprocessed, r.facets, r.collisions, r.census))
    assert np.array_equal(np.array(events), z["events"])          # exact integers
    got = sim.particle_arrays()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(got[f], z[f]), f
    for f in ("x", "y", "energy", "weight", "dt_to_census"):
        scale = np.maximum(np.abs(z[f]), 1e-300)
        assert np.max(np.abs(got[f] - z[f]) / scale) < 1e-9, f   # ocml vs glibc log/sincos
    for f in ("omega_x", "omega_y"):
        assert np.max(np.abs(got[f] - z[f])) < 1e-9, f
    t = sim.tally_host()
    assert np.linalg.norm(t - z["tally"]) / np.linalg.norm(z["tally"]) < 1e-9   # bar: 1e-6
    sim.close()
