"""The arithmetic policy of the event bodies on a real MI355X (`-m gpu`).

Every history kernel exists twice in libneutral_hip.so: instantiated with the bare
division / square-root / logarithm sequences (proven on densities and table entries
in [2^-100, 2^100]) and with IEEE-checked arithmetic (whatever the reference's C
accepts).  The library picks per step, on the device, from the step's own density
mesh and tables (include/neutral_hip.h: neutral_hip_set_arithmetic).  Checked here
against the CPU oracle, which is plain C like omp3/neutral.c:127-146,231,311-317 and
therefore runs a true-vacuum cell (density 0: cell_mfp = 1/0 = inf) on infinities:
event counts exact, cells and death flags exact, floating state and the per-cell
tally to 1e-9 -- the bars of tests/test_hip_parity.py -- with no rebuild flag."""
import numpy as np
import pytest

import oracle_binding as ob
from conftest import gpu_available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]

TALLY_L2_TOL = 1e-9
STATE_TOL = 1e-9


@pytest.fixture()
def iface():
    from neutral_amd import interface
    interface.set_quiet(True)
    interface.set_lazy_export(False)
    interface.set_arithmetic(interface.ARITH_AUTO)
    yield interface
    interface.set_arithmetic(interface.ARITH_AUTO)


def _close(got, want, tol, absolute=False):
    """equal where the oracle is not finite (same infinity / both NaN), within tol elsewhere"""
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])
    d = np.abs(got[fin] - want[fin])
    s = 1.0 if absolute else np.maximum(np.abs(want[fin]), 1e-300)
    assert d.size == 0 or float(np.max(d / s)) < tol


def _against_oracle(iface, prob, cs, variant, steps, cs_absorb=None):
    sim = iface.Simulation(prob, *cs, variant=variant, cs_absorb=cs_absorb)
    ref = ob.OracleRun(prob, *cs, cs_absorb=cs_absorb)
    sim.inject()
    ref.inject()
    stats = []
    for tt in range(1, steps + 1):
        g, c = sim.step(tt), ref.step(tt)
        assert (g.nprocessed, g.facets, g.collisions, g.census) == \
            (c.nprocessed, c.facets, c.collisions, c.census), tt
        stats.append((g.stats.checked_arithmetic, g.stats.attempts))
    gp, cp = sim.particle_arrays(), ref.particles.as_dict()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(gp[f], cp[f]), f
    for f in ("energy", "weight", "dt_to_census", "x", "y"):
        _close(gp[f], cp[f], STATE_TOL)
    for f in ("omega_x", "omega_y"):
        _close(gp[f], cp[f], STATE_TOL, absolute=True)
    fin = np.isfinite(cp["mfp_to_collision"])
    scale = max(1e-300, float(np.max(np.abs(cp["mfp_to_collision"][fin]), initial=0.0)))
    _close(gp["mfp_to_collision"] / scale, cp["mfp_to_collision"] / scale, STATE_TOL, absolute=True)
    tg, tc = sim.tally_host(), ref.tally
    assert np.all(np.isfinite(tc)) and np.all(np.isfinite(tg))
    assert np.linalg.norm(tg - tc) / np.linalg.norm(tc) < TALLY_L2_TOL
    assert np.array_equal(tg == 0.0, tc == 0.0)
    sim.close()
    return stats, cp


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_true_vacuum_runs_as_the_reference_does(iface, make_problem, cs, variant):
    """csp with its vacuum at density 0 instead of 1e-30 (omp3/neutral.c:131-135: cell_mfp =
    1/0 = inf, every event a facet, deposition 0; a free flight drawn in vacuum is inf mean
    free paths long and survives into the dense block).  The first step's fast attempt is
    turned down on the device and runs again checked; later steps start checked."""
    prob = make_problem("csp", nx=100, nparticles=20000, iterations=4, dt=1.0e-6)
    prob.density[prob.density < 1.0e-20] = 0.0
    assert (prob.density == 0.0).any() and (prob.density > 1.0).any()
    stats, cp = _against_oracle(iface, prob, cs, variant, 4)
    assert stats[0] == (1, 2), stats            # turned down once, then checked
    assert all(s == (1, 1) for s in stats[1:]), stats
    assert np.isinf(cp["mfp_to_collision"]).any()   # the case tests what it says


@pytest.mark.parametrize("variant", [0, 2])
def test_extreme_densities_run_as_the_reference_does(iface, make_problem, cs, variant):
    """A vacuum of 1e-250 and a block of 1e40: macroscopic cross sections of 5e-250 and 5e40,
    free paths that overflow to inf or are 1e-41 long -- outside what the fast sequences
    are proven on, inside what IEEE arithmetic (and the reference) computes."""
    prob = make_problem("csp", nx=64, nparticles=8192, iterations=3, dt=1.0e-6)
    dense = prob.density > 1.0
    prob.density[~dense] = 1.0e-250
    prob.density[dense] = 1.0e40
    stats, _ = _against_oracle(iface, prob, cs, variant, 3)
    assert stats[0] == (1, 2) and stats[-1] == (1, 1), stats


def test_table_entries_outside_the_proven_range(iface, make_problem, cs):
    """An absorb table that is zero above 100 eV (no absorption there: p_absorb = 0 / Sigma_s
    = 0; below, histories are absorbed and die as usual, so energies stay inside the keys) is
    input the reference accepts; zeros are outside [2^-100, 2^100], so the steps run checked."""
    keys, values = cs
    absorb = (keys.copy(), np.where(keys > 100.0, 0.0, values))
    prob = make_problem("csp", nx=64, nparticles=8192, iterations=3, dt=1.0e-6)
    stats, _ = _against_oracle(iface, prob, cs, 2, 3, cs_absorb=absorb)
    assert stats[0] == (1, 2) and stats[1] == (1, 1), stats


@pytest.mark.parametrize("deck,nx,n,its,dt", [("csp", 100, 20000, 3, 1.0e-6),
                                              ("split", 128, 20000, 2, None),
                                              ("stream", 100, 20000, 2, None)])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_checked_kernels_give_the_fast_kernels_bits(iface, make_problem, cs, deck, nx, n, its, dt,
                                                    variant):
    """On input inside the proven range the two instantiations are the same function: every
    particle field bit for bit, every event count."""
    kw = dict(nx=nx, nparticles=n, iterations=its)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    out = {}
    for mode in (iface.ARITH_AUTO, iface.ARITH_CHECKED):
        iface.set_arithmetic(mode)
        sim = iface.Simulation(prob, *cs, variant=variant)
        sim.inject()
        ev = []
        for tt in range(1, its + 1):
            r = sim.step(tt)
            ev.append((r.nprocessed, r.facets, r.collisions, r.census))
            assert r.stats.checked_arithmetic == (1 if mode == iface.ARITH_CHECKED else 0)
            assert r.stats.attempts == 1
        out[mode] = (sim.particle_arrays(), sim.tally_host(), ev)
        sim.close()
    fast, chk = out[iface.ARITH_AUTO], out[iface.ARITH_CHECKED]
    assert fast[2] == chk[2]
    for f in fast[0]:
        assert np.array_equal(fast[0][f], chk[0][f]), f
    assert np.linalg.norm(fast[1] - chk[1]) / np.linalg.norm(fast[1]) < 1e-13


def test_policy_follows_the_input_step_by_step(iface, make_problem, cs):
    """Auto mode decides per step: a cell of true vacuum written into the density mesh between
    two steps makes the next step run checked (one attempt turned down), taking it out again
    brings the fast kernels back -- and the particles are those of the oracle throughout."""
    prob = make_problem("stream", nx=64, nparticles=4096, iterations=5)
    sim = iface.Simulation(prob, *cs, variant=2)
    ref = ob.OracleRun(prob, *cs)
    sim.inject()
    ref.inject()
    cell = 33 * 64 + 31   # inside the source box: histories start in it and cross it
    seen = []
    for tt, rho in ((1, None), (2, 0.0), (3, 0.0), (4, 1.0e-30), (5, None)):
        if rho is not None:
            sim.density[cell] = rho
            ref.density[cell] = rho
        g, c = sim.step(tt), ref.step(tt)
        assert (g.nprocessed, g.facets, g.collisions) == (c.nprocessed, c.facets, c.collisions)
        seen.append((g.stats.checked_arithmetic, g.stats.attempts))
    # (step 4 still starts checked -- what step 3's check found -- and learns that it need not)
    assert seen == [(0, 1), (1, 2), (1, 1), (1, 1), (0, 1)], seen
    tg, tc = sim.tally_host(), ref.tally
    assert np.linalg.norm(tg - tc) / np.linalg.norm(tc) < TALLY_L2_TOL
    sim.close()
