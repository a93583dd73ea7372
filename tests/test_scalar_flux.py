"""Scalar-flux tally (neutral_data.h:95: declared by the reference, written by none of
its backends).  The definition this repository implements -- the path-length
estimator sum(weight * segment length) / N per cell, flushed where the energy
deposition is -- is stated in oracle/neutral_oracle.c; there is no reference output
to compare with, so the tests are (1) what the definition implies, on the oracle
(CPU) and on the HIP path (GPU), and (2) HIP path == oracle."""
import numpy as np
import pytest

import oracle_binding as ob
from conftest import gpu_available

EV_TO_J = 1.60217646e-19          # neutral_data.h:17
PARTICLE_MASS = 1.674927471213e-27  # neutral_data.h:20


def _speed(energy_ev):
    return np.sqrt(2.0 * energy_ev * EV_TO_J / PARTICLE_MASS)   # omp3/neutral.c:117


def test_oracle_flux_in_a_collision_free_deck(make_problem, cs):
    """stream deck: uniform near-vacuum, nobody collides, every particle keeps weight 1
    and energy E0.  Then (a) the energy tally is the flux times ONE constant (the
    heating factor of omp3/neutral.c:481-494 at E0), cell by cell, and (b) the flux
    sums to the distance every particle travels: speed * dt."""
    prob = make_problem("stream", nx=100, nparticles=3000, iterations=2)
    run = ob.OracleRun(prob, *cs, scalar_flux=True)
    run.inject()
    for tt in (1, 2):
        r = run.step(tt)
        assert r.collisions == 0 and r.census == 3000
    touched = run.flux > 0
    assert np.array_equal(touched, run.tally > 0)
    ratio = run.tally[touched] / run.flux[touched]
    assert (ratio.max() - ratio.min()) / ratio.mean() < 1e-12
    assert run.flux.sum() == pytest.approx(2 * _speed(prob.initial_energy) * prob.dt, rel=1e-12)


def test_oracle_flux_does_not_disturb_the_energy_tally(make_problem, cs):
    prob = make_problem("csp", nx=64, nparticles=4000, iterations=2, dt=2.0e-6)
    a = ob.OracleRun(prob, *cs)
    b = ob.OracleRun(prob, *cs, scalar_flux=True)
    a.inject()
    b.inject()
    for tt in (1, 2):
        ra, rb = a.step(tt), b.step(tt)
        assert (ra.facets, ra.collisions, ra.nprocessed) == (rb.facets, rb.collisions, rb.nprocessed)
    # (same histories; the OpenMP atomics add in whatever order the threads arrive)
    assert np.linalg.norm(a.tally - b.tally) / np.linalg.norm(a.tally) < 1e-13
    # absorptions halve weights: the flux of a collided history is below its path length
    assert 0 < b.flux.sum() < 2 * _speed(prob.initial_energy) * prob.dt


gpu = pytest.mark.gpu
needs_gpu = pytest.mark.skipif(not gpu_available(), reason="needs a GPU")


@pytest.fixture()
def iface():
    from neutral_amd import interface
    interface.set_quiet(True)
    interface.set_lazy_export(False)
    interface.set_variant(interface.VARIANT_OVER_PARTICLE)
    return interface


CASES = [
    # deck, nx, nparticles, iterations, dt
    ("stream", 100, 20000, 2, None),
    ("csp", 100, 30000, 3, 1.0e-6),
    ("split", 128, 20000, 2, None),
    ("scatter", 64, 4096, 2, None),
    ("csp", 37, 1000, 2, 3.0e-6),
]


@gpu
@needs_gpu
@pytest.mark.parametrize("deck,nx,n,its,dt", CASES)
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_flux_matches_oracle(iface, make_problem, cs, monkeypatch, deck, nx, n, its, dt, variant):
    kw = dict(nx=nx, nparticles=n, iterations=its)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "32")   # small decks under windows too
    ref = ob.OracleRun(prob, *cs, scalar_flux=True)
    sim = iface.Simulation(prob, *cs, variant=variant, scalar_flux=True)
    plain = iface.Simulation(prob, *cs, variant=variant)
    ref.inject()
    sim.inject()
    plain.inject()
    for tt in range(1, its + 1):
        c, g, p = ref.step(tt), sim.step(tt), plain.step(tt)
        assert (g.nprocessed, g.facets, g.collisions) == (c.nprocessed, c.facets, c.collisions)
        assert (g.nprocessed, g.facets, g.collisions) == (p.nprocessed, p.facets, p.collisions)
    flux = sim.flux.cpu().numpy()
    assert np.linalg.norm(flux - ref.flux) / np.linalg.norm(ref.flux) < 1e-9
    assert abs(flux.sum() - ref.flux.sum()) / ref.flux.sum() < 1e-10
    assert np.array_equal(flux == 0.0, ref.flux == 0.0)
    # keeping the flux changes neither the histories nor (beyond summation order) the energy tally
    a, b = sim.particle_arrays(), plain.particle_arrays()
    for f in a:
        assert np.array_equal(a[f], b[f]), f
    te, tp = sim.tally_host(), plain.tally_host()
    assert np.linalg.norm(te - tp) / np.linalg.norm(tp) < 1e-13
    assert np.linalg.norm(te - ref.tally) / np.linalg.norm(ref.tally) < 1e-9
    sim.close()
    plain.close()


@gpu
@needs_gpu
def test_flux_survives_the_time_sliced_collision_stage(iface, make_problem, cs, monkeypatch):
    """A history set aside in the middle of its collision chain carries pending
    weight * path length along (it is flushed only at the next facet, census or death)."""
    prob = make_problem("csp", nx=100, nparticles=100000, iterations=2, dt=1.0e-6)
    monkeypatch.setenv("NEUTRAL_K2_MAX_BLOCKS", "4")
    sim = iface.Simulation(prob, *cs, variant=2, scalar_flux=True)
    ref = ob.OracleRun(prob, *cs, scalar_flux=True)
    sim.inject()
    ref.inject()
    requeued = 0
    for tt in (1, 2):
        g, c = sim.step(tt), ref.step(tt)
        requeued += g.stats.requeued
        assert (g.facets, g.collisions) == (c.facets, c.collisions)
    assert requeued > 0
    flux = sim.flux.cpu().numpy()
    assert np.linalg.norm(flux - ref.flux) / np.linalg.norm(ref.flux) < 1e-9
    sim.close()


@gpu
@needs_gpu
@pytest.mark.parametrize("tile", [16, 32, 64, 128])
def test_flux_windows_at_every_tile_edge(iface, make_problem, cs, monkeypatch, tile):
    """Two 88-cell windows share the LDS when the flux is kept, so tiles stop at 64
    cells (a request for 128 is served with the densest-fitting choice)."""
    prob = make_problem("stream", nx=400, nparticles=30000, iterations=1)
    monkeypatch.setenv("NEUTRAL_TILE_CELLS", str(tile))
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "32")
    sim = iface.Simulation(prob, *cs, variant=2, scalar_flux=True)
    sim.inject()
    r = sim.step(1)
    assert r.stats.tile_cells == (tile if tile <= 64 else 64)
    assert r.stats.stream_passes > 1      # histories did change windows
    flux, tally = sim.flux.cpu().numpy(), sim.tally_host()
    touched = flux > 0
    assert np.array_equal(touched, tally > 0)
    ratio = tally[touched] / flux[touched]
    assert (ratio.max() - ratio.min()) / ratio.mean() < 1e-9
    assert flux.sum() == pytest.approx(_speed(prob.initial_energy) * prob.dt, rel=1e-10)
    sim.close()


@gpu
@needs_gpu
def test_flux_properties_at_the_stream_config_full_size(iface, make_problem, cs):
    """BASELINE config 2 (stream 400^2, 1e7 particles) with the flux kept: the two
    size-independent properties of a collision-free deck."""
    prob = make_problem("stream", nx=400, nparticles=10_000_000, iterations=1)
    sim = iface.Simulation(prob, *cs, variant=2, scalar_flux=True)
    sim.inject()
    r = sim.step(1)
    assert r.collisions == 0 and r.census == 10_000_000
    flux, tally = sim.flux.cpu().numpy(), sim.tally_host()
    touched = flux > 0
    assert np.array_equal(touched, tally > 0)
    ratio = tally[touched] / flux[touched]
    assert (ratio.max() - ratio.min()) / ratio.mean() < 1e-9
    assert flux.sum() == pytest.approx(_speed(prob.initial_energy) * prob.dt, rel=1e-10)
    sim.close()
