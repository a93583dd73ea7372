"""Parity of the HIP path (through the C-ABI) with the CPU oracle, on a real
MI355X.  Run with `-m gpu`.

Bars (BASELINE.json north_star: "tallies within 1e-6 relative, Philox/Threefry
stream reproduced bit-exact"):
  * Threefry words and the (0,1] doubles: bit-exact;
  * cross-section bracket index: exact; value: bit-exact (same IEEE operations);
  * integer particle state (cellx, celly, dead) and event counts: exact;
  * floating particle state: 1e-9 relative (log/sincos come from ocml on the
    GPU and glibc on the CPU and differ in the last ulp; everything else is
    IEEE add/mul/div/sqrt in the same order);
  * per-cell tally: relative L2 <= 1e-9, global sum <= 1e-10 (north-star 1e-6).
"""
import numpy as np
import pytest

import oracle_binding as ob
from conftest import gpu_available

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]

TALLY_L2_TOL = 1e-9       # north-star bar: 1e-6
TALLY_SUM_TOL = 1e-10
STATE_TOL = 1e-9


@pytest.fixture(scope="module")
def _iface_module():
    from neutral_amd import interface
    interface.set_quiet(True)
    return interface


@pytest.fixture()
def iface(_iface_module):
    # the kernel variant is process-global state of the library: every test
    # starts from the default (over-particle) unless it asks for another one
    _iface_module.set_variant(_iface_module.VARIANT_OVER_PARTICLE)
    return _iface_module


def _rel(a, b):
    d = np.abs(a - b)
    s = np.maximum(np.abs(b), 1e-300)
    return float(np.max(d / s)) if a.size else 0.0


# ---- unit known answers on the device ------------------------------------------

def test_threefry_bit_exact(iface, pins):
    rows = [[int(v["ctr"][0], 16), int(v["key"][0], 16), int(v["key"][1], 16)]
            for v in pins["threefry2x64_20"] if int(v["ctr"][1], 16) == 0]
    rng = np.random.default_rng(11)
    rnd = rng.integers(0, 2**64, size=(50000, 3), dtype=np.uint64)
    edge = np.array([[c, p, m] for c in (0, 1, 2**64 - 1) for p in (0, 2**32, 2**64 - 1)
                     for m in (0, 1, 2**63)], dtype=np.uint64)
    inp = np.vstack([np.array(rows, dtype=np.uint64), edge, rnd])
    words, rn = iface.probe_threefry(inp)
    for i, row in enumerate(inp):
        if i < len(rows) + len(edge) or i % 25 == 0:
            c, p, m = (int(x) for x in row)
            assert (int(words[i, 0]), int(words[i, 1])) == ob.threefry(c, 0, p, m)
            assert (rn[i, 0], rn[i, 1]) == ob.generate_random_numbers(p, m, c)
    for v, w in zip([v for v in pins["threefry2x64_20"] if int(v["ctr"][1], 16) == 0], words):
        assert (int(w[0]), int(w[1])) == (int(v["out"][0], 16), int(v["out"][1], 16))
    assert rn.min() > 0.0 and rn.max() <= 1.0


def test_cs_lookup_matches_oracle(iface, pins, cs):
    import torch
    keys, values = cs
    dk = torch.from_numpy(keys).cuda()
    dv = torch.from_numpy(values).cuda()
    table = iface.CrossSection(dk.data_ptr(), dv.data_ptr(), len(keys))
    rng = np.random.default_rng(5)
    es = np.concatenate([
        np.array([e["energy"] for e in pins["cs_lookup"]]),
        keys[:-1][::37], np.nextafter(keys[1:][::41], 0.0),
        np.exp(rng.uniform(np.log(keys[0] * 1.0001), np.log(keys[-1] * 0.9999), 100000))])
    value, index = iface.probe_cs_lookup(table, es, use_index=True)
    value_b, index_b = iface.probe_cs_lookup(table, es, use_index=False)
    # the exponent-bucketed search and the plain bisection find the same bracket
    assert np.array_equal(index, index_b) and np.array_equal(value, value_b)
    host = ob.CsTable(keys, values)
    for i in list(range(len(pins["cs_lookup"]))) + list(range(5, len(es), 97)):
        v, ind = host.lookup(float(es[i]))
        assert index[i] == ind
        assert value[i] == v
    for e, v, ind in zip(pins["cs_lookup"], value, index):
        assert ind == e["index"] and v == pytest.approx(e["value"], rel=1e-14)
    # whole-array check of the bracket against numpy
    assert np.array_equal(index, np.searchsorted(keys, es, side="right") - 1)


def test_cs_lookup_index_on_awkward_tables(iface):
    """Bucketed index on tables unlike the shipped one: tiny, clustered keys,
    many binades, keys on bucket boundaries (powers of two)."""
    import torch
    rng = np.random.default_rng(9)
    tables = [
        np.array([1.0, 2.0]),
        np.array([0.5, 1.0, 2.0, 4.0, 8.0, 1024.0]),
        np.sort(np.concatenate([1.0 + rng.random(3000) * 1e-9, [0.25, 7.0]])),
        np.sort(np.exp(rng.uniform(np.log(1e-300), np.log(1e300), 5000))),
        np.cumsum(rng.random(60000)) + 1e-3,
    ]
    for keys in tables:
        keys = np.unique(keys)
        values = rng.random(keys.size) * 100.0
        dk, dv = torch.from_numpy(keys).cuda(), torch.from_numpy(values).cuda()
        table = iface.CrossSection(dk.data_ptr(), dv.data_ptr(), keys.size)
        es = np.concatenate([keys[:-1], np.nextafter(keys[1:], 0.0),
                             rng.uniform(keys[0], keys[-1], 20000)])
        es = es[(es >= keys[0]) & (es < keys[-1])]
        v1, i1 = iface.probe_cs_lookup(table, es, use_index=True)
        v0, i0 = iface.probe_cs_lookup(table, es, use_index=False)
        assert np.array_equal(i1, np.searchsorted(keys, es, side="right") - 1)
        assert np.array_equal(i1, i0) and np.array_equal(v1, v0)


def test_distance_to_facet_matches_oracle(iface):
    rng = np.random.default_rng(3)
    n = 20000
    ex_lo = rng.uniform(0, 0.9, n)
    ey_lo = rng.uniform(0, 0.9, n)
    h = 1.0 / 400
    theta = rng.uniform(0, 2 * np.pi, n)
    rows = np.stack([ex_lo + rng.uniform(0, h, n), ey_lo + rng.uniform(0, h, n),
                     np.cos(theta), np.sin(theta), rng.uniform(1e4, 2e7, n),
                     ex_lo, ex_lo + h, ey_lo, ey_lo + h], axis=1)
    # axis-aligned directions and a particle sitting on an edge
    rows[0, 2:4] = (1.0, 0.0)
    rows[1, 2:4] = (0.0, -1.0)
    rows[2, 0] = rows[2, 5]
    dist, xf = iface.probe_distance_to_facet(rows)
    import ctypes as C
    for i in list(range(3)) + list(range(3, n, 53)):
        r = rows[i]
        edgex = np.array([r[5], r[6]])
        edgey = np.array([r[7], r[8]])
        d, x = C.c_double(), C.c_int()
        ob.lib().orc_calc_distance_to_facet(
            r[0], r[1], 0, 0, 0, r[2], r[3], r[4], 0, 0, C.byref(d), C.byref(x),
            edgex.ctypes.data_as(C.POINTER(C.c_double)),
            edgey.ctypes.data_as(C.POINTER(C.c_double)))
        assert xf[i] == x.value
        assert dist[i] == pytest.approx(d.value, rel=1e-15) or (np.isinf(d.value) and np.isinf(dist[i]))


# ---- injection -------------------------------------------------------------------

def test_quotient_through_kept_reciprocal_is_the_ieee_quotient(iface):
    """The stream kernel divides a facet's path length by speed and mean free path
    through reciprocals it keeps across facets, and the fast arithmetic policy divides
    that way everywhere (neutral_device.h: refined_reciprocal, quotient_by_reciprocal).
    The device's own `a / b` is the correctly rounded quotient numpy computes.  The
    quotient through the reciprocal -- one Newton step on v_rcp_f64 and one residual
    correction since round 5 -- is the same bits for every pair of operands with
    ordinary mantissas (four million here; 8.6e9 in tools/micro/one_step.hip), and
    within one ulp where BOTH mantissas are of the adversarial kind (2 - 2^-k: quotients
    that sit within 2^-97 of a rounding boundary, which the second Newton step of the
    compiler's sequence resolves and one step does not)."""
    rng = np.random.default_rng(20260203)
    n = 2_000_000
    def doubles(lo_exp, hi_exp, count):
        mant = rng.random(count) + 1.0                       # [1, 2)
        # some mantissas with long runs of ones / zeros: the hard cases of division
        hard = rng.integers(0, 4, count) == 0
        bits = rng.integers(0, 52, count)
        mant_hard = 2.0 - np.ldexp(1.0, -bits)
        mant = np.where(hard, mant_hard, mant)
        sign = np.where(rng.integers(0, 8, count) == 0, -1.0, 1.0)
        return sign * np.ldexp(mant, rng.integers(lo_exp, hi_exp, count)), hard
    parts_a = [doubles(-299, 299, n), doubles(-25, -5, n), doubles(-340, 340, n // 4)]
    parts_b = [doubles(-299, 299, n), doubles(8, 100, n), doubles(-340, 340, n // 4)]
    a = np.concatenate([p[0] for p in parts_a])
    b = np.concatenate([p[0] for p in parts_b])
    both_hard = np.concatenate([p[1] for p in parts_a]) & np.concatenate([p[1] for p in parts_b])
    q_dev, q_kept, plain = iface.probe_division(np.stack([a, b], axis=1))
    with np.errstate(all="ignore"):
        q_np = a / b
    assert np.array_equal(q_dev.view(np.uint64), q_np.view(np.uint64))
    assert plain[: 2 * n].all() and not plain.all()
    ordinary = plain & ~both_hard
    assert ordinary.sum() > 3_500_000
    assert np.array_equal(q_kept[ordinary].view(np.uint64), q_dev[ordinary].view(np.uint64))
    adversarial = plain & both_hard
    step = np.abs(q_kept[adversarial].view(np.int64) - q_dev[adversarial].view(np.int64))
    assert int(step.max()) <= 1
    print("quotient through the one-step reciprocal: %d of %d adversarial pairs one ulp off, 0 of %d ordinary ones"
          % (int((step != 0).sum()), int(adversarial.sum()), int(ordinary.sum())))


def test_a_scatter_seeded_from_its_neighbours_gives_the_ieee_results(iface):
    """The fast arithmetic policy works out a scatter's second quotient and second root from the
    first, the speed after it from the speed before it, and the direction's two reciprocals off one
    (neutral_device.h: scatter_cosine, speed_after_scatter; neutral_history.h:
    refresh_direction_plain_or_wrapped) -- one residual correction each after a seed that is
    good to 2^-46 and better.  Two million random scatters over the tables' energy range, with
    nearly axis-parallel directions among them: the bits of the evaluation with IEEE divisions
    and roots (the checked policy's), which are what numpy computes from omp3/neutral.c:263-265,
    :297 and :435-436 (the speed and the reciprocals bit for bit).  (tools/micro/scatter_cosine.hip counts 8.6e9 on the device: none differ.)"""
    rng = np.random.default_rng(2026)
    n = 2_000_000
    e = 1.0e-2 * 2.0 ** (rng.random(n) * 33.2)                 # 1e-2 ... 1e8 eV
    mu = 1.0 - 2.0 * rng.random(n)
    ang = 2.0 * np.pi * rng.random(n)
    ox, oy = np.cos(ang), np.sin(ang)
    thin = rng.integers(0, 64, n) == 0
    ox = np.where(thin, ox * 2.0 ** -rng.integers(1, 140, n), ox)     # some nearly axis-parallel
    got = iface.probe_scatter(np.stack([e, mu, ox, oy], axis=1))
    same = lambda a, b: np.array_equal(a.view(np.uint64), b.view(np.uint64))
    e_new = got["e_new"]
    assert np.all((e_new > 0.95 * e) & (e_new < 1.05 * e))
    mass_no = 100.0
    with np.errstate(all="ignore"):
        cos_np = 0.5 * ((mass_no + 1.0) * np.sqrt(e_new / e) - (mass_no - 1.0) * np.sqrt(e / e_new))
        speed_np = np.sqrt((2.0 * e_new * 1.60217646e-19) / 1.674927471213e-27)
        ux_np, uy_np = 1.0 / (ox * speed_np), 1.0 / (oy * speed_np)
    # (the cosine's last combination a * s1 - b * s2 may be contracted into a fused multiply-add on the
    #  device, as in the reference's own -O3 build: numpy's two roundings are an ulp or two away)
    assert np.max(np.abs(got["cos_ieee"] - cos_np)) < 4.0e-14
    assert same(got["speed_ieee"], speed_np)
    assert same(got["u_x_inv_ieee"], ux_np) and same(got["u_y_inv_ieee"], uy_np)
    assert same(got["cos_fast"], got["cos_ieee"])
    assert same(got["speed_fast"], got["speed_ieee"])
    assert same(got["u_x_inv_fast"], got["u_x_inv_ieee"]) and same(got["u_y_inv_fast"], got["u_y_inv_ieee"])


def test_log_of_a_sample_is_faithful(iface):
    """The history kernels take -log of a (0,1] sample with their own evaluation
    (neutral_device.h: log_of_sample) instead of the device library's.  Measured here
    against an 80-bit reference: below one ulp everywhere (a faithful result, what
    libm promises), correct at the ends of the sample range, and it agrees with the
    device library to the last bit or its neighbour."""
    rng = np.random.default_rng(42)
    n = 2_000_000
    x = np.concatenate([
        rng.random(n),                                   # the bulk of the samples
        1.0 - rng.random(n // 4) * 2.0 ** -rng.integers(1, 52, n // 4),   # just below one
        np.ldexp(rng.random(n // 4) + 1.0, -rng.integers(1, 66, n // 4).astype(np.int32)),
        np.array([1.0, 0.5, 2.0 ** -65, 2.0 ** -64 + 2.0 ** -65, np.nextafter(1.0, 0.0),
                  0.70710678118654746, 0.70710678118654757, 3.0, 1e300, 2.5e-308]),
    ])
    x = x[(x > 0.0)]
    mine, lib = iface.probe_log(x)
    ref = np.log(x.astype(np.longdouble))
    ulp = np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)
    ulp[ulp == 0] = np.longdouble(5e-324)
    err = np.abs(mine.astype(np.longdouble) - ref) / ulp
    assert float(err.max()) < 1.0, float(err.max())
    assert np.all(mine[x == 1.0] == 0.0) and (x == 1.0).any()
    # the library's result is the same double or the next one
    step = np.abs(mine.view(np.int64) - lib.view(np.int64))
    assert int(step.max()) <= 1
    print("log_of_sample: max error %.3f ulp, %.2f %% differ from the library by one ulp"
          % (float(err.max()), 100.0 * float((step == 1).mean())))


def test_quotients_by_the_two_constants_are_ieee_quotients(iface):
    """x / PARTICLE_MASS (speed, omp3/neutral.c:297) and x / (MASS_NO+1)^2 (energy after a
    scatter, :252) go through the correctly rounded reciprocal of the constant and one
    residual correction (neutral_device.h: quotient_by_constant): the bits of the IEEE
    quotient, for numerators the path produces and for everything else."""
    rng = np.random.default_rng(77)
    n = 2_000_000
    mant = rng.random(n) + 1.0
    hard = np.where(rng.integers(0, 3, n) == 0, 2.0 - np.ldexp(1.0, -rng.integers(0, 52, n)), mant)
    x = np.concatenate([
        np.ldexp(hard, rng.integers(-299, 299, n).astype(np.int32)),
        rng.random(n // 2) * 3.2e-13,                    # 2 E eV_TO_J for E up to 1 MeV
        10.0 ** rng.uniform(-2, 8, n // 2) * 10201.0,    # E (A^2 + 2 A mu + 1)
        np.ldexp(mant[: n // 8], rng.integers(-1070, 1023, n // 8).astype(np.int32)),
        np.array([0.0, -0.0, 1.0, np.inf, 5e-324, -3.0]),
    ])
    m_fast, m_dev, a_fast, a_dev = iface.probe_constant_quotients(x)
    with np.errstate(all="ignore"):
        m_ref = x / 1.674927471213e-27
        a_ref = x / 10201.0
    for got, ref in ((m_dev, m_ref), (m_fast, m_ref), (a_dev, a_ref), (a_fast, a_ref)):
        assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))


def test_square_root_without_the_wrapping_is_the_ieee_root(iface):
    """neutral_device.h: sqrt_plain_range keeps the compiler's ten-operation core of
    the f64 square root and leaves its range scaling out for arguments in
    [2^-500, 2^500]: the bits must be numpy's (correctly rounded) everywhere, inside
    the range, outside it and for the special values."""
    rng = np.random.default_rng(9)
    n = 2_000_000
    mant = rng.random(n) + 1.0
    hard = np.where(rng.integers(0, 3, n) == 0, 2.0 - np.ldexp(1.0, -rng.integers(0, 52, n)), mant)
    x = np.concatenate([
        np.ldexp(hard, rng.integers(-499, 499, n).astype(np.int32)),
        1.0 - rng.random(n // 2) ** 2,                   # sin^2 of a scattering angle
        1.0 + rng.random(n // 2) * 0.05,                 # energy ratios
        np.ldexp(mant[: n // 4], rng.integers(-1070, 1023, n // 4).astype(np.int32)),
        np.array([0.0, -0.0, 1.0, 4.0, np.inf, 5e-324, 2.0 ** -500, 2.0 ** 500, 1e-320]),
    ])
    mine, dev = iface.probe_sqrt(x)
    ref = np.sqrt(x)
    assert np.array_equal(dev.view(np.uint64), ref.view(np.uint64))
    assert np.array_equal(mine.view(np.uint64), ref.view(np.uint64))
    neg = iface.probe_sqrt(np.array([-1.0, np.nan]))[0]
    assert np.isnan(neg).all()


@pytest.mark.parametrize("deck", ["scatter", "stream", "csp", "split"])
def test_inject_matches_oracle(iface, make_problem, cs, deck):
    prob = make_problem(deck, nx=100, nparticles=30000, iterations=1)
    sim = iface.Simulation(prob, *cs)
    ref = ob.OracleRun(prob, *cs)
    sim.inject()
    ref.inject()
    g, c = sim.particle_arrays(), ref.particles.as_dict()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(g[f], c[f]), f
    for f in ("energy", "weight", "dt_to_census", "mfp_to_collision"):
        assert np.array_equal(g[f], c[f]), f
    for f in ("x", "y"):
        assert _rel(g[f], c[f]) < 1e-15, f
    for f in ("omega_x", "omega_y"):
        assert np.max(np.abs(g[f] - c[f])) < 1e-15, f
    sim.close()


# ---- the history loop --------------------------------------------------------------

CASES = [
    # deck, nx, nparticles, iterations, dt
    ("scatter", 64, 4096, 2, None),
    ("stream", 100, 20000, 2, None),
    ("csp", 100, 20000, 3, 1.0e-6),
    ("split", 128, 20000, 2, None),
    ("csp", 37, 1000, 2, 3.0e-6),     # ragged: odd mesh, n not a multiple of 64/256
    ("scatter", 16, 1, 1, None),      # a single particle
    ("split", 400, 63, 1, 2.0e-6),
]


@pytest.mark.parametrize("deck,nx,n,its,dt", CASES)
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_history_matches_oracle(iface, make_problem, cs, deck, nx, n, its, dt, variant):
    kw = dict(nx=nx, nparticles=n, iterations=its)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    sim = iface.Simulation(prob, *cs, variant=variant)
    ref = ob.OracleRun(prob, *cs)
    sim.inject()
    ref.inject()
    for tt in range(1, its + 1):
        g, c = sim.step(tt), ref.step(tt)
        assert (g.nprocessed, g.facets, g.collisions) == (c.nprocessed, c.facets, c.collisions)
    gp, cp = sim.particle_arrays(), ref.particles.as_dict()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(gp[f], cp[f]), f
    for f in ("energy", "weight", "dt_to_census", "x", "y"):
        assert _rel(gp[f], cp[f]) < STATE_TOL, f
    for f in ("omega_x", "omega_y"):
        assert np.max(np.abs(gp[f] - cp[f])) < STATE_TOL, f
    # mfp_to_collision is a difference of like quantities: absolute on its scale
    scale = max(1e-300, float(np.max(np.abs(cp["mfp_to_collision"]))))
    assert np.max(np.abs(gp["mfp_to_collision"] - cp["mfp_to_collision"])) / scale < STATE_TOL
    tg, tc = sim.tally_host(), ref.tally
    assert np.linalg.norm(tg - tc) / np.linalg.norm(tc) < TALLY_L2_TOL
    assert abs(tg.sum() - tc.sum()) / abs(tc.sum()) < TALLY_SUM_TOL
    # cells the oracle never touched stay exactly zero
    assert np.array_equal(tg == 0.0, tc == 0.0)
    sim.close()


@pytest.mark.parametrize("variant", [0, 2])
def test_uneven_mesh_matches_oracle(iface, make_problem, cs, variant):
    """Edges that are NOT the host layer's formula edge[i] = dx * i (cells of growing width):
    the device-side check of the mesh formula fails, the stream kernel loads its edges, and
    the histories are the oracle's -- as on the uniform meshes, where it works them out."""
    prob = make_problem("csp", nx=96, nparticles=20000, iterations=3, dt=1.0e-6)
    t = np.linspace(0.0, 1.0, prob.nx + 1)
    prob.edgex[:] = prob.edgex[-1] * (0.7 * t + 0.3 * t * t)   # monotone, uneven
    prob.edgey[:] = prob.edgey[-1] * (0.8 * t + 0.2 * t ** 3)
    prob.edgedx[:-1] = np.diff(prob.edgex)
    prob.edgedy[:-1] = np.diff(prob.edgey)
    sim = iface.Simulation(prob, *cs, variant=variant)
    ref = ob.OracleRun(prob, *cs)
    sim.inject()
    ref.inject()
    for tt in (1, 2, 3):
        g, c = sim.step(tt), ref.step(tt)
        assert (g.nprocessed, g.facets, g.collisions) == (c.nprocessed, c.facets, c.collisions)
    gp, cp = sim.particle_arrays(), ref.particles.as_dict()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(gp[f], cp[f]), f
    for f in ("energy", "weight", "dt_to_census", "x", "y"):
        assert _rel(gp[f], cp[f]) < STATE_TOL, f
    tg, tc = sim.tally_host(), ref.tally
    assert np.linalg.norm(tg - tc) / np.linalg.norm(tc) < TALLY_L2_TOL
    assert np.array_equal(tg == 0.0, tc == 0.0)
    sim.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_distinct_tables_take_the_two_search_path(iface, make_problem, cs, variant):
    keys, values = cs
    absorb = (keys.copy(), values * 0.5)
    prob = make_problem("csp", nx=64, nparticles=8192, iterations=2, dt=2.0e-6)
    sim = iface.Simulation(prob, keys, values, cs_absorb=absorb, variant=variant)
    ref = ob.OracleRun(prob, keys, values, cs_absorb=absorb)
    sim.inject()
    ref.inject()
    for tt in (1, 2):
        g, c = sim.step(tt), ref.step(tt)
        assert iface.last_step().same_tables == 0
        assert (g.nprocessed, g.facets, g.collisions) == (c.nprocessed, c.facets, c.collisions)
    tg, tc = sim.tally_host(), ref.tally
    assert np.linalg.norm(tg - tc) / np.linalg.norm(tc) < TALLY_L2_TOL
    sim.close()


def test_identical_tables_detected(iface, make_problem, cs):
    prob = make_problem("scatter", nx=32, nparticles=512, iterations=1)
    sim = iface.Simulation(prob, *cs)
    sim.inject()
    sim.step(1)
    assert iface.last_step().same_tables == 1
    sim.close()


def test_out_of_particles_and_dead_particles_are_skipped(iface, make_problem, cs, capfd):
    # scatter: every particle dies in step 1, so step 2
    # processes nothing and leaves the tally untouched
    prob = make_problem("scatter", nx=32, nparticles=2048, iterations=2)
    sim = iface.Simulation(prob, *cs)
    sim.inject()
    r1 = sim.step(1)
    t1 = sim.tally_host().copy()
    r2 = sim.step(2)
    assert r1.nprocessed == 2048 and r2.nprocessed == 0 and r2.collisions == 0
    assert np.array_equal(t1, sim.tally_host())
    assert np.all(sim.particle_arrays()["dead"] == 1)
    # zero local particles: the reference prints and returns (omp3/neutral.c:30-33)
    import ctypes as C
    iface.set_quiet(False)
    sim.nlocal = C.c_int(0)
    sim.step(3)
    iface.set_quiet(True)
    iface.library().neutral_hip_synchronize()
    out = capfd.readouterr().out
    assert "Out of particles" in out
    sim.close()


def test_particle_shards_reproduce_the_unsharded_run(iface, make_problem, cs):
    """any contiguous partition of the particle ids gives the same
    histories; the sum of the shard tallies equals the one-GPU tally."""
    prob = make_problem("csp", nx=64, nparticles=10000, iterations=2, dt=2.0e-6)
    whole = iface.Simulation(prob, *cs)
    whole.inject()
    res = [whole.step(tt) for tt in (1, 2)]
    t_whole = whole.tally_host()
    p_whole = whole.particle_arrays()
    whole.close()
    bounds = [0, 3333, 3334, 10000]
    t_sum = np.zeros_like(t_whole)
    facets = collisions = 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        sh = iface.Simulation(prob, *cs, shard=(a, b - a))
        sh.inject()
        for tt in (1, 2):
            r = sh.step(tt)
            facets += r.facets
            collisions += r.collisions
        t_sum += sh.tally_host()
        ps = sh.particle_arrays()
        for f in ("cellx", "celly", "dead", "energy", "x"):
            assert np.array_equal(ps[f], p_whole[f][a:b]), f
        sh.close()
    assert facets == sum(r.facets for r in res)
    assert collisions == sum(r.collisions for r in res)
    assert np.linalg.norm(t_sum - t_whole) / np.linalg.norm(t_whole) < 1e-12


def test_reinject_resets_state(iface, make_problem, cs):
    prob = make_problem("split", nx=64, nparticles=5000, iterations=1, dt=1.0e-6)
    sim = iface.Simulation(prob, *cs)
    sim.inject()
    r1 = sim.step(1)
    t1 = sim.tally_host().copy()
    sim.inject()          # reset, no allocation
    sim.zero_tally()
    r2 = sim.step(1)
    assert (r1.facets, r1.collisions) == (r2.facets, r2.collisions)
    assert np.linalg.norm(sim.tally_host() - t1) / np.linalg.norm(t1) < 1e-12
    sim.close()


# ---- the reference's own known answers through validate() --------------------------

@pytest.mark.parametrize("variant", [0, 2])
@pytest.mark.parametrize("name", ["stream", "csp", "scatter"])
def test_reference_known_answers_default_decks(iface, make_problem, cs, name, variant, tmp_path,
                                               capfd):
    """problems/neutral.tests:1-3 at the decks' default sizes (4000^2 mesh), checked
    by the library's validate() exactly as main.c:154 does."""
    from neutral_amd import decks
    d = decks.STANDARD_DECKS[name]
    prob = make_problem(name)  # defaults
    assert (prob.nx, prob.nparticles, prob.niters) == (d["nx"], d["nparticles"], d["iterations"])
    tests_file = decks.write_tests_file(str(tmp_path / "neutral.tests"), {name: prob.deck})
    iface.set_tests_file(tests_file)
    sim = iface.Simulation(prob, *cs, variant=variant)
    sim.inject()
    for tt in range(1, prob.niters + 1):
        sim.step(tt)
    sim.validate()
    iface.library().neutral_hip_synchronize()
    out = capfd.readouterr().out
    assert "PASSED validation." in out, out
    total = float(sim.tally_host().sum())
    assert abs(total - decks.KNOWN_ANSWERS[name]) / decks.KNOWN_ANSWERS[name] < 1e-3
    if name == "stream":
        import closed_form  # (the deck as shipped against its closed form: 1e-12)
        exact = closed_form.stream_deck_tally(*cs)
        assert abs(total - exact) / exact < 1e-12
    sim.close()


# ---- BASELINE sizes: size-independent properties -------------------------------------

@pytest.mark.parametrize("variant", [0, 1, 2])
def test_stream_400_matches_recorded_omp3_counts(iface, make_problem, cs, pins, variant):
    """BASELINE config 2 shape at 1e6 particles: exact facet count of the omp3 run
    recorded in BASELINE.md, tally to 1e-9."""
    r = pins["omp3_runs"][1]
    prob = make_problem("stream", nx=r["nx"], nparticles=r["nparticles"], iterations=1)
    sim = iface.Simulation(prob, *cs, variant=variant)
    sim.inject()
    s = sim.step(1)
    assert (s.facets, s.collisions) == (r["facets"], r["collisions"])
    assert float(sim.tally_host().sum()) == pytest.approx(r["tally"], rel=1e-9)
    sim.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_split_800_matches_recorded_omp3_counts(iface, make_problem, cs, pins, variant):
    r = pins["omp3_runs"][4]
    prob = make_problem("split", nx=r["nx"], nparticles=r["nparticles"], iterations=1)
    sim = iface.Simulation(prob, *cs, variant=variant)
    sim.inject()
    s = sim.step(1)
    assert (s.facets, s.collisions) == (r["facets"], r["collisions"])
    assert float(sim.tally_host().sum()) == pytest.approx(r["tally"], rel=1e-9)
    sim.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_csp_400_ten_steps_matches_recorded_omp3(iface, make_problem, cs, pins, variant):
    r = pins["omp3_runs"][3]
    prob = make_problem("csp", nx=r["nx"], nparticles=r["nparticles"], iterations=10)
    sim = iface.Simulation(prob, *cs, variant=variant)
    sim.inject()
    for tt in range(1, 11):
        s = sim.step(tt)
    assert s.nprocessed == r["last_step_processed"]
    assert float(sim.tally_host().sum()) == pytest.approx(r["tally"], rel=1e-9)
    sim.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_stream_deck_closed_form(iface, make_problem, cs, variant):
    """The stream deck against its closed form (tests/closed_form.py: reference-held constants,
    the shipped table, omp3/neutral.c:117,474-495 -- no oracle in between): the global tally is
    speed * dt * sigma_t * BARNS * heating * n per timestep whatever the mesh and the particle
    count.  HIP path at 1e-12 (problems/neutral.tests:2 holds 1e-3), every variant, two steps."""
    import closed_form
    prob = make_problem("stream", nx=200, nparticles=200000, iterations=2)
    sim = iface.Simulation(prob, *cs, variant=variant)
    sim.inject()
    for tt in (1, 2):
        s = sim.step(tt)
        assert s.collisions == 0 and s.nprocessed == 200000
        exact = closed_form.stream_deck_tally(*cs, iterations=tt)
        assert abs(float(sim.tally_host().sum()) - exact) / exact < 1e-12
    sim.close()


def test_stream_tally_is_intensive_in_particle_count(iface, make_problem, cs):
    """stream: uniform near-vacuum, no collisions, so the global tally per source
    particle is independent of N -- checked at 1e7 particles
    (BASELINE config 2) and at 1e5 against the deck's closed form (tests/closed_form.py) at
    1e-12, and against the shipped known answer at the reference's own 1e-3."""
    import closed_form
    from neutral_amd import decks
    exact = closed_form.stream_deck_tally(*cs)
    for n in (100000, 10000000):
        prob = make_problem("stream", nx=400, nparticles=n, iterations=1)
        sim = iface.Simulation(prob, *cs, variant=2)
        sim.inject()
        s = sim.step(1)
        assert s.collisions == 0 and s.nprocessed == n
        total = float(sim.tally_host().sum())
        sim.close()
        assert abs(total - exact) / exact < 1e-12, (n, total, exact)
        assert abs(total - decks.KNOWN_ANSWERS["stream"]) / decks.KNOWN_ANSWERS["stream"] < 1e-3


@pytest.mark.parametrize("deck,nx,n,dt", [("csp", 100, 50000, 1.0e-6), ("split", 200, 60000, 5.0e-7),
                                          ("stream", 400, 30000, None), ("scatter", 64, 20000, None)])
def test_kernel_variants_are_bitwise_identical_in_particle_state(iface, make_problem, cs, deck, nx,
                                                                 n, dt):
    """K2/K3 run the same event bodies with the same RNG counters as K1, so particle
    end states agree bit for bit; tallies differ only by summation order.  The
    stream case makes particles outrun the LDS window (multi-pass migration)."""
    kw = dict(nx=nx, nparticles=n, iterations=3)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    out = []
    for variant in (0, 1, 2):
        sim = iface.Simulation(prob, *cs, variant=variant)
        sim.inject()
        ev = [sim.step(tt) for tt in (1, 2, 3)]
        assert iface.last_step().variant == variant
        out.append((sim.particle_arrays(), sim.tally_host(),
                    [(r.nprocessed, r.facets, r.collisions, r.census) for r in ev]))
        sim.close()
    p0, t0, e0 = out[0]
    for p1, t1, e1 in out[1:]:
        assert e0 == e1
        for f in p0:
            assert np.array_equal(p0[f], p1[f]), f
        assert np.linalg.norm(t0 - t1) / np.linalg.norm(t0) < 1e-13


@pytest.mark.parametrize("deck,nx,n,dt,its,blocks", [("scatter", 64, 40000, None, 1, 32),
                                                     ("csp", 100, 100000, 1.0e-6, 2, 4),
                                                     ("split", 200, 100000, 2.0e-7, 2, 32)])
def test_collision_stage_time_slicing_is_bitwise_neutral(iface, make_problem, cs, monkeypatch,
                                                         deck, nx, n, dt, its, blocks):
    """With more queued colliders than lanes the collision stage round-robins each
    wave's share through an LDS ring (histories are suspended mid-chain and resumed,
    possibly on another lane).  A grid of a few workgroups makes that happen at a size the
    oracle finishes in seconds: same particle bits as the over-particle kernel, same
    event counts as the oracle, and the swaps really took place."""
    kw = dict(nx=nx, nparticles=n, iterations=its)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    monkeypatch.setenv("NEUTRAL_K2_MAX_BLOCKS", str(blocks))
    sim = iface.Simulation(prob, *cs, variant=2)
    sim.inject()
    ev2, requeued, suspended = [], 0, 0
    for tt in range(1, its + 1):
        r = sim.step(tt)
        ev2.append((r.nprocessed, r.facets, r.collisions, r.census))
        requeued += iface.last_step().requeued
        suspended += iface.last_step().suspended
        assert iface.last_step().aborted == 0
    p2, t2 = sim.particle_arrays(), sim.tally_host()
    sim.close()
    monkeypatch.delenv("NEUTRAL_K2_MAX_BLOCKS")
    assert requeued > 0, f"the case no longer exercises time slicing ({suspended} colliders)"

    sim = iface.Simulation(prob, *cs, variant=0)
    sim.inject()
    ev0 = []
    for tt in range(1, its + 1):
        r = sim.step(tt)
        ev0.append((r.nprocessed, r.facets, r.collisions, r.census))
    p0, t0 = sim.particle_arrays(), sim.tally_host()
    sim.close()
    assert ev0 == ev2
    for f in p0:
        assert np.array_equal(p0[f], p2[f]), f
    assert np.linalg.norm(t0 - t2) / np.linalg.norm(t0) < 1e-13

    ref = ob.OracleRun(prob, *cs)
    ref.inject()
    for tt in range(1, its + 1):
        c = ref.step(tt)
        assert (c.nprocessed, c.facets, c.collisions) == ev2[tt - 1][:3]
    assert np.linalg.norm(t2 - ref.tally) / np.linalg.norm(ref.tally) < TALLY_L2_TOL


def _random_deck_text(rng):
    """A deck the reference's loader would accept: a source box, a vacuum background
    and one to three denser boxes, on a mesh that need not be square."""
    nx, ny = int(rng.integers(20, 260)), int(rng.integers(20, 260))
    sw, sh = rng.uniform(0.02, 0.5, 2)
    sx, sy = rng.uniform(0.0, 1.0 - sw), rng.uniform(0.0, 1.0 - sh)
    lines = [f"source xpos={sx!r} ypos={sy!r} width={sw!r} height={sh!r}",
             "problem_0 density=1e-30 energy=0.0 xpos=0.0 ypos=0.0 width=1.0 height=1.0"]
    for i in range(int(rng.integers(1, 4))):
        bw, bh = rng.uniform(0.05, 0.7, 2)
        bx, by = rng.uniform(0.0, 1.0 - bw), rng.uniform(0.0, 1.0 - bh)
        dens = float(10.0 ** rng.uniform(-2, 4))
        lines.append(f"problem_{i + 1} density={dens!r} energy=1.0 xpos={bx!r} ypos={by!r} "
                     f"width={bw!r} height={bh!r}")
    n = int(rng.integers(1, 20000))
    e0 = float(10.0 ** rng.uniform(2, 6))
    dt = float(10.0 ** rng.uniform(-8, -6.3))
    its = int(rng.integers(1, 4))
    lines += [f"nparticles {n}", f"initial_energy {e0!r}", f"dt {dt!r}", f"nx {nx}", f"ny {ny}",
              f"iterations {its}", "visit_dump 0"]
    return "\n".join(lines) + "\n", its


@pytest.mark.parametrize("seed", range(64))
def test_random_decks_match_oracle(iface, cs, tmp_path, monkeypatch, seed):
    """Differential run on decks nobody tuned for: random source and density boxes,
    non-square meshes, random particle counts, energies and timesteps, a random kernel
    variant, and (for the tiled one) a random small grid of the collision kernel so
    that pooled shares of every size, time slicing included, are exercised.  Event
    counts and integer state exact, floating-point state and tallies to rounding."""
    from neutral_amd import decks, host
    rng = np.random.default_rng(7000 + seed)
    text, its = _random_deck_text(rng)
    path = tmp_path / "random.params"
    path.write_text(text)
    prob = host.setup_problem(str(path), decks.ARCH_WIDTH, decks.ARCH_HEIGHT)
    variant = int(rng.integers(0, 3))
    blocks = [None, 1, 3, 16][int(rng.integers(0, 4))]
    if blocks is not None:
        monkeypatch.setenv("NEUTRAL_K2_MAX_BLOCKS", str(blocks))
    sim = iface.Simulation(prob, *cs, variant=variant)
    ref = ob.OracleRun(prob, *cs)
    sim.inject()
    ref.inject()
    for tt in range(1, its + 1):
        g, c = sim.step(tt), ref.step(tt)
        assert (g.nprocessed, g.facets, g.collisions, g.census) == \
            (c.nprocessed, c.facets, c.collisions, c.census), text
        assert iface.last_step().aborted == 0
    gp, cp = sim.particle_arrays(), ref.particles.as_dict()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(gp[f], cp[f]), f
    for f in ("energy", "weight", "dt_to_census", "x", "y"):
        assert _rel(gp[f], cp[f]) < STATE_TOL, f
    for f in ("omega_x", "omega_y"):
        assert np.max(np.abs(gp[f] - cp[f])) < STATE_TOL, f
    tg, tc = sim.tally_host(), ref.tally
    if np.linalg.norm(tc) > 0.0:
        assert np.linalg.norm(tg - tc) / np.linalg.norm(tc) < TALLY_L2_TOL
    assert np.array_equal(tg == 0.0, tc == 0.0)
    sim.close()


def test_tiled_variant_lazy_export_and_variant_switches(iface, make_problem, cs):
    """The tiled variant's private record store and the SoA arrays stay coherent:
    lazy export + explicit sync, switching variants between steps, reinjection."""
    prob = make_problem("csp", nx=100, nparticles=30000, iterations=4, dt=1.0e-6)
    ref = iface.Simulation(prob, *cs, variant=0)
    ref.inject()
    for tt in (1, 2, 3, 4):
        ref.step(tt)
    want, t_want = ref.particle_arrays(), ref.tally_host()
    ref.close()

    iface.set_lazy_export(True)
    try:
        sim = iface.Simulation(prob, *cs, variant=2)
        sim.inject()
        sim.step(1)                      # tiled, records ahead of the SoA arrays
        sim.step(2)                      # tiled again: records re-sorted in place
        iface.set_variant(1)
        sim.step(3)                      # K2 needs the SoA arrays: implicit sync
        iface.set_variant(2)
        sim.step(4)                      # re-import
        got = sim.particle_arrays()      # explicit sync inside
        for f in want:
            assert np.array_equal(got[f], want[f]), f
        assert np.linalg.norm(sim.tally_host() - t_want) / np.linalg.norm(t_want) < 1e-13
        # reinjection invalidates the records
        sim.inject()
        sim.zero_tally()
        r = sim.step(1)
        ref2 = iface.Simulation(prob, *cs, variant=0)
        ref2.inject()
        r0 = ref2.step(1)
        assert (r.facets, r.collisions) == (r0.facets, r0.collisions)
        a, b = sim.particle_arrays(), ref2.particle_arrays()
        for f in a:
            assert np.array_equal(a[f], b[f]), f
        sim.close()
        ref2.close()
    finally:
        iface.set_lazy_export(False)


def test_full_size_csp_step_properties(iface, make_problem, cs):
    """BASELINE config 4 at its full size on one GPU (csp 400^2, 1e8 particles), two
    timesteps of the default (tiled) pipeline: size-independent properties --
    every source particle is processed, nobody collides before reaching the block
    (the source box and the dense block are 0.1 apart, a particle moves 0.138 per
    step), the tally is non-negative, touches only cells a particle can have
    reached, and is intensive in N (equal to the 1e6-particle tally per unit N to
    Monte-Carlo accuracy)."""
    totals = {}
    for n in (1000000, 100000000):
        prob = make_problem("csp", nx=400, nparticles=n, iterations=2)
        sim = iface.Simulation(prob, *cs, variant=2)
        iface.set_lazy_export(True)
        try:
            sim.inject()
            r1 = sim.step(1)
            r2 = sim.step(2)
        finally:
            iface.set_lazy_export(False)
        assert r1.nprocessed == n and r2.nprocessed == n
        assert r1.collisions == 0 and r1.census == n
        assert r1.stats.stream_passes == 1
        t = sim.tally_host().reshape(400, 400)
        assert np.all(t >= 0.0)
        # after 2 steps nothing can be further than 2*0.138+ from the source box [0.1,0.3]^2
        assert np.all(t[:, 240:] == 0.0) and np.all(t[240:, :] == 0.0)
        totals[n] = (float(t.sum()), (r1.facets + r2.facets) / n)
        sim.close()
    # (collisions in the block during step 2 dominate the tally: ~1 % Monte-Carlo noise at 1e6)
    assert totals[1000000][0] == pytest.approx(totals[100000000][0], rel=3e-2)
    assert totals[1000000][1] == pytest.approx(totals[100000000][1], rel=2e-3)


@pytest.mark.parametrize("deck,nx", [("scatter", 400), ("split", 800)])
def test_full_size_collision_decks_are_intensive(iface, make_problem, cs, deck, nx):
    """BASELINE configs 3 and 5 at full size on one GPU (1e8 particles: the collision
    queue runs 30 000 histories per wave through the first-come path): every source
    particle is processed exactly once, the tally is non-negative and confined to
    where particles can be, and events per particle and the (N-normalised) tally
    equal the 1e6-particle run to Monte-Carlo accuracy.  On scatter every history
    ends in its first timestep (E0 = 1 keV, ~700 collisions), so a second timestep
    finds nobody."""
    per = {}
    for n in (1000000, 100000000):
        prob = make_problem(deck, nx=nx, nparticles=n, iterations=2)
        sim = iface.Simulation(prob, *cs, variant=2)
        iface.set_lazy_export(True)
        try:
            sim.inject()
            r1 = sim.step(1)
            assert iface.last_step().aborted == 0
            r2 = sim.step(2) if deck == "scatter" else None
        finally:
            iface.set_lazy_export(False)
        assert r1.nprocessed == n
        t = sim.tally_host().reshape(nx, nx)
        assert np.all(t >= 0.0) and np.isfinite(t).all()
        if deck == "scatter":
            assert r2.nprocessed == 0 and r2.collisions == 0
            assert r1.census == 0
            # mean free path 1e-5: nothing leaves the source box [0.2, 0.8]^2 by more than a cell
            lo, hi = int(0.2 * nx) - 2, int(0.8 * nx) + 2
            assert np.all(t[:lo, :] == 0.0) and np.all(t[hi:, :] == 0.0)
            assert np.all(t[:, :lo] == 0.0) and np.all(t[:, hi:] == 0.0)
        per[n] = (r1.collisions / n, r1.facets / n, float(t.sum()))
        sim.close()
    small, big = per[1000000], per[100000000]
    assert small[0] == pytest.approx(big[0], rel=2e-3)
    assert small[2] == pytest.approx(big[2], rel=5e-3)
    if deck == "split":
        assert small[1] == pytest.approx(big[1], rel=5e-3)


@pytest.mark.parametrize("deck,nx,its", [("csp", 400, 10), ("split", 800, 1), ("scatter", 400, 1),
                                         ("stream", 400, 1)])
def test_baseline_shapes_per_cell_tally_l2(iface, make_problem, cs, deck, nx, its):
    """The BASELINE.json configurations at their mesh sizes with 1e6 particles,
    default (tiled) pipeline against the oracle: exact per-step event counts and
    the per-cell tally L2 ("tally L2 vs omp3"), bar 1e-6, asserted at 1e-9."""
    prob = make_problem(deck, nx=nx, nparticles=1000000, iterations=its)
    sim = iface.Simulation(prob, *cs, variant=2)
    ref = ob.OracleRun(prob, *cs)
    sim.inject()
    ref.inject()
    for tt in range(1, its + 1):
        g, c = sim.step(tt), ref.step(tt)
        assert (g.nprocessed, g.facets, g.collisions, g.census) == \
            (c.nprocessed, c.facets, c.collisions, c.census)
    tg, tc = sim.tally_host(), ref.tally
    assert np.linalg.norm(tg - tc) / np.linalg.norm(tc) < TALLY_L2_TOL
    assert np.array_equal(tg == 0.0, tc == 0.0)
    gp, cp = sim.particle_arrays(), ref.particles.as_dict()
    for f in ("cellx", "celly", "dead"):
        assert np.array_equal(gp[f], cp[f]), f
    sim.close()
