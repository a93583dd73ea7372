"""The tiled pipeline's own machinery, on a real MI355X (`-m gpu`): the counting
sort and the tile edge chosen per problem, stream passes enqueued without host
round trips, the SoA arrays kept current by the kernels (and re-imported when a
caller rewrites them), the cached view of the cross-section tables.

Everything here is checked against the over-particle kernel (variant 0), which
test_hip_parity.py checks against the CPU oracle: same particle bits, same event
counts, tallies equal up to summation order."""
import ctypes as C

import numpy as np
import pytest

from conftest import gpu_available

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]


@pytest.fixture()
def iface():
    from neutral_amd import interface
    interface.set_quiet(True)
    interface.set_lazy_export(False)
    interface.set_variant(interface.VARIANT_OVER_PARTICLE)
    return interface


def _run(iface, prob, cs, variant, steps, between=None, **sim_kw):
    sim = iface.Simulation(prob, *cs, variant=variant, **sim_kw)
    sim.inject()
    ev, stats = [], []
    for tt in range(1, steps + 1):
        if between is not None:
            between(sim, tt)
        r = sim.step(tt)
        ev.append((r.nprocessed, r.facets, r.collisions, r.census))
        stats.append(r.stats)
    out = (sim.particle_arrays(), sim.tally_host(), ev, stats)
    sim.close()
    return out


def _loaded_hip_runtime():
    """The HIP runtime this process already uses (torch's), by its mapped path."""
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime mapped")


def _same(a, b):
    pa, ta, ea, _ = a
    pb, tb, eb, _ = b
    assert ea == eb
    for f in pa:
        assert np.array_equal(pa[f], pb[f]), f
    assert np.linalg.norm(ta - tb) / np.linalg.norm(ta) < 1e-13


@pytest.mark.parametrize("deck,nx,n,dt,steps", [
    ("stream", 400, 30000, None, 2),     # long flights: many windows per history
    ("csp", 100, 50000, 1.0e-6, 3),      # vacuum + dense block
    ("split", 200, 60000, 5.0e-7, 2),
    ("stream", 1000, 20000, None, 1),    # sparse: 0.02 particles per cell
])
@pytest.mark.parametrize("tile", [16, 32, 64, 128])
def test_every_tile_edge_gives_the_same_histories(iface, make_problem, cs, monkeypatch, deck, nx,
                                                  n, dt, steps, tile):
    """The tile edge (16..128 cells under the 128-cell window) only decides where a
    tally is accumulated and how often a history changes windows.  A low window
    threshold makes the small test problems stream under windows at all."""
    kw = dict(nx=nx, nparticles=n, iterations=steps)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    want = _run(iface, prob, cs, 0, steps)
    monkeypatch.setenv("NEUTRAL_TILE_CELLS", str(tile))
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "32")
    got = _run(iface, prob, cs, 2, steps)
    _same(want, got)
    assert all(s.tile_cells == tile for s in got[3])
    assert all(s.aborted == 0 for s in got[3])


def test_meshes_with_more_tiles_than_the_sort_keeps_in_lds(iface, make_problem, cs, monkeypatch):
    """1600^2 cells at a forced tile edge of 16 are 10 000 tiles: the counting sort places
    with one global atomic per record instead of its LDS histogram (8 191 tiles at most)."""
    prob = make_problem("stream", nx=1600, nparticles=20000, iterations=1)
    want = _run(iface, prob, cs, 0, 1)
    monkeypatch.setenv("NEUTRAL_TILE_CELLS", "16")
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "8")
    got = _run(iface, prob, cs, 2, 1)
    _same(want, got)
    assert got[3][0].tile_cells == 16


@pytest.mark.parametrize("deck,nx,n,dt,tile", [
    ("stream", 2048, 3000, None, 16),   # 16 384 tiles, a few hundred histories per late pass
    ("csp", 1600, 6000, 2.0e-6, 32),    # 2 500 tiles; colliders thin the passes out further
    ("stream", 4000, 900, None, 128),   # the shipped mesh with fewer particles than tiles
])
def test_thin_passes_over_many_tiles(iface, make_problem, cs, monkeypatch, deck, nx, n, dt, tile):
    """Late passes of a sparse deck hold a few hundred histories spread over thousands of
    tiles: the chunk list then comes from threads that each sweep a bounded run of tiles
    (tile_chunks_kernel), as un-windowed chunks much smaller than a chunk.  Same histories
    as the over-particle kernel, bit for bit."""
    kw = dict(nx=nx, nparticles=n, iterations=1)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    want = _run(iface, prob, cs, 0, 1)
    monkeypatch.setenv("NEUTRAL_TILE_CELLS", str(tile))
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "8")
    got = _run(iface, prob, cs, 2, 1)
    _same(want, got)
    assert got[3][0].tile_cells == tile
    assert got[3][0].aborted == 0


@pytest.mark.parametrize("deck,nx,n,dt,steps", [
    ("split", 200, 1000000, 5.0e-7, 2),   # every particle collides: 240 histories per wave
    ("scatter", 200, 800000, None, 1),
])
def test_histories_taken_over_by_another_wave(iface, make_problem, cs, monkeypatch, deck, nx, n, dt,
                                              steps):
    """A wave of the collision stage that has emptied its ring takes half of what waits in the
    ring of a wave of its own CU (history_regroup_kernel: try_steal) -- the oldest wave of a
    SIMD gets most of its issue slots and is through its share long before the others.  By
    default only rings of hundreds of histories are taken from (1e8-particle runs);
    NEUTRAL_STEAL_MIN=1 lets the 240 a million-particle deck gives every wave be taken too.
    Which wave finishes a history changes nothing it computes: same bits as the
    over-particle kernel, same event counts -- with and without."""
    kw = dict(nx=nx, nparticles=n, iterations=steps)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    want = _run(iface, prob, cs, 0, steps)
    monkeypatch.setenv("NEUTRAL_STEAL_MIN", "1")
    got = _run(iface, prob, cs, 2, steps)
    _same(want, got)
    assert sum(s.steals for s in got[3]) > 100
    assert all(s.aborted == 0 for s in got[3])
    monkeypatch.setenv("NEUTRAL_STEAL_MIN", "0")
    alone = _run(iface, prob, cs, 2, steps)
    _same(want, alone)
    assert sum(s.steals for s in alone[3]) == 0


@pytest.mark.parametrize("deck,nx,n,dt,steps,weight", [
    ("split", 200, 1000000, 5.0e-7, 2, 5),    # 244 histories per wave: 610 / 122 / 122 / 122
    ("scatter", 200, 800000, None, 1, 3),     # 195 per wave, a last round that is partial
    ("scatter", 128, 530001, None, 1, 8),     # the smallest full grid: 129 per wave, 8 : 1 : 1 : 1
])
def test_weighted_shares_at_test_size(iface, make_problem, cs, monkeypatch, deck, nx, n, dt, steps,
                                      weight):
    """Shares in proportion to what a wave is served (history_regroup_kernel: ring_map /
    ring_index, 5 : 1 : 1 : 1 over the four waves of a SIMD) are dealt by default only from 256
    histories per wave on -- 1e7-particle runs, out of the default suite's reach (round-4 advisor
    finding).  NEUTRAL_WEIGHTED_SHARE_MIN lowers that bar, NEUTRAL_STEAL_MIN=1 lets thieves use the
    victims' weighted maps on rings of any size: a full grid of 4 096 waves, partial last rounds,
    wrap-arounds of the small rings -- same bits as the over-particle kernel, no history lost or
    run twice (the event counts say so), and the stage reports that the weighted map was the
    one in use."""
    kw = dict(nx=nx, nparticles=n, iterations=steps)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    want = _run(iface, prob, cs, 0, steps)
    monkeypatch.setenv("NEUTRAL_WEIGHTED_SHARE_MIN", "8")
    monkeypatch.setenv("NEUTRAL_SHARE_WEIGHT", str(weight))
    monkeypatch.setenv("NEUTRAL_STEAL_MIN", "1")
    got = _run(iface, prob, cs, 2, steps)
    _same(want, got)
    assert max(s.weighted_waves for s in got[3]) >= 4000, [s.weighted_waves for s in got[3]]
    assert sum(s.steals for s in got[3]) > 100
    assert sum(s.requeued for s in got[3]) > 0
    assert all(s.aborted == 0 and s.steals_refused == 0 for s in got[3]), \
        [(s.aborted, s.steals_refused, s.steals, s.weighted_waves, s.suspended) for s in got[3]]
    monkeypatch.setenv("NEUTRAL_SHARE_WEIGHT", "1")
    equal = _run(iface, prob, cs, 2, steps)
    _same(want, equal)
    assert all(s.weighted_waves == 0 for s in equal[3])


def test_a_slow_thief_is_waited_for(iface, make_problem, cs, monkeypatch):
    """The owner of a ring does not store into it while a thief is still copying what it took
    (round-3 advisor finding: only timing kept the two apart).  NEUTRAL_STEAL_DELAY makes every
    thief sleep between its take and its copy -- far longer than the owner's next time slice --
    with the smallest rings that are taken from (NEUTRAL_STEAL_MIN=1) on a small grid, so that
    the owners' hand-backs wrap round to the places the thief has yet to read.  Same bits as
    the over-particle kernel, no history lost or run twice (the event counts say so)."""
    prob = make_problem("split", nx=200, nparticles=300000, iterations=1, dt=5.0e-7)
    want = _run(iface, prob, cs, 0, 1)
    monkeypatch.setenv("NEUTRAL_STEAL_MIN", "1")
    monkeypatch.setenv("NEUTRAL_STEAL_DELAY", "400")   # x 64 sleep units: tens of microseconds
    monkeypatch.setenv("NEUTRAL_K2_MAX_BLOCKS", "512")
    got = _run(iface, prob, cs, 2, 1)
    _same(want, got)
    assert sum(s.steals for s in got[3]) > 50
    assert sum(s.requeued for s in got[3]) > 0
    assert all(s.aborted == 0 and s.steals_refused == 0 for s in got[3]), \
        [(s.aborted, s.steals_refused, s.steals, s.weighted_waves, s.suspended) for s in got[3]]


def test_two_stores_stepped_in_turn_on_two_streams(iface, make_problem, cs, monkeypatch):
    """The words the collision stage's stealing works on belong to the tiled workspace and are
    reset by a kernel on the stream of the launch that uses them (they were process-wide
    symbols reset through pointers cached in function statics).  Two particle stores stepped
    in turn, each on a stream of its own, with stealing on: each ends with the bits the
    over-particle kernel gives it."""
    import torch
    a_prob = make_problem("split", nx=200, nparticles=1000000, iterations=2, dt=5.0e-7)
    b_prob = make_problem("csp", nx=128, nparticles=300000, iterations=2, dt=1.0e-6)
    want_a = _run(iface, a_prob, cs, 0, 2)
    want_b = _run(iface, b_prob, cs, 0, 2)
    monkeypatch.setenv("NEUTRAL_STEAL_MIN", "1")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    a = iface.Simulation(a_prob, *cs, variant=2)
    b = iface.Simulation(b_prob, *cs, variant=2)
    a.inject()
    b.inject()
    torch.cuda.synchronize()
    ev_a, ev_b, steals = [], [], 0
    for tt in (1, 2):
        iface.set_stream(sa.cuda_stream)
        r = a.step(tt)
        ev_a.append((r.nprocessed, r.facets, r.collisions, r.census))
        steals += r.stats.steals
        iface.set_stream(sb.cuda_stream)
        r = b.step(tt)
        ev_b.append((r.nprocessed, r.facets, r.collisions, r.census))
        steals += r.stats.steals
    iface.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got_a = (a.particle_arrays(), a.tally_host(), ev_a, None)
    got_b = (b.particle_arrays(), b.tally_host(), ev_b, None)
    a.close()
    b.close()
    _same(want_a, got_a)
    _same(want_b, got_b)
    assert steals > 0


@pytest.mark.parametrize("deck,nx,n,dt,steps,tile", [
    ("stream", 400, 200000, None, 2, 16),    # every history crosses ~9 windows per step
    ("stream", 1000, 100000, None, 1, 128),  # tile = window: every crossing of a tile edge is a hop
    ("csp", 128, 200000, 4.0e-6, 3, 16),     # long steps: flights outrun the window, then collide
])
def test_migrants_change_tiles_inside_the_launch(iface, make_problem, cs, monkeypatch, deck, nx, n, dt,
                                                 steps, tile):
    """The asynchronous tile queue (neutral_hip_set_stream_queues / NEUTRAL_STREAM_QUEUES=1): a
    history that leaves its tally window with far to go is
    handed, inside the stream kernel, to the queue of the tile it has reached, and whichever
    workgroup claims it streams it on under a window centred there -- no sort, no further pass.
    Same bits as the over-particle kernel; the step takes ONE stream pass where the pass
    mechanism (the default) takes several, with the same bits again; and a queue too
    small for what a tile receives (NEUTRAL_STREAM_QUEUE_CAPACITY) overflows into passes, same
    bits again."""
    kw = dict(nx=nx, nparticles=n, iterations=steps)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    want = _run(iface, prob, cs, 0, steps)
    monkeypatch.setenv("NEUTRAL_TILE_CELLS", str(tile))
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "32")
    monkeypatch.setenv("NEUTRAL_STREAM_QUEUES", "1")
    got = _run(iface, prob, cs, 2, steps)
    _same(want, got)
    assert all(s.aborted == 0 for s in got[3])
    assert sum(s.stream_hops for s in got[3]) > n // 4
    assert all(s.stream_overflows == 0 for s in got[3])
    assert all(s.stream_passes == 1 for s in got[3])
    monkeypatch.setenv("NEUTRAL_STREAM_QUEUES", "0")
    passes = _run(iface, prob, cs, 2, steps)
    _same(want, passes)
    assert all(s.stream_hops == 0 for s in passes[3])
    assert max(s.stream_passes for s in passes[3]) > 1
    monkeypatch.setenv("NEUTRAL_STREAM_QUEUES", "1")
    monkeypatch.setenv("NEUTRAL_STREAM_QUEUE_CAPACITY", "64")
    tight = _run(iface, prob, cs, 2, steps)
    _same(want, tight)
    assert sum(s.stream_overflows for s in tight[3]) > 0
    assert sum(s.stream_hops for s in tight[3]) > 0
    assert all(s.aborted == 0 for s in tight[3])


def test_tile_edge_follows_the_particle_density(iface, make_problem, cs):
    for deck, nx, n, want in (("csp", 100, 100000, 16), ("csp", 200, 100000, 32),
                              ("csp", 400, 100000, 64), ("stream", 1000, 20000, 128)):
        prob = make_problem(deck, nx=nx, nparticles=n, iterations=1)
        sim = iface.Simulation(prob, *cs, variant=2)
        sim.inject()
        assert sim.step(1).stats.tile_cells == want, (deck, nx, n)
        sim.close()


def test_a_steady_state_step_waits_for_the_device_once(iface, make_problem, cs, monkeypatch):
    """From the second step of a problem on, stream passes, collision queue and
    collision stage are enqueued on what the step before needed: the only wait is the
    read-back of the counters.  The stream deck makes every history migrate through
    several windows: through several passes per step (the default), or inside ONE launch of the
    stream kernel with the tile queues (NEUTRAL_STREAM_QUEUES=1)."""
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "32")
    prob = make_problem("stream", nx=400, nparticles=30000, iterations=4)
    want = _run(iface, prob, cs, 0, 4)
    monkeypatch.setenv("NEUTRAL_STREAM_QUEUES", "1")
    got = _run(iface, prob, cs, 2, 4)
    _same(want, got)
    assert all(s.stream_passes == 1 and s.stream_hops > 30000 for s in got[3])
    assert all(s.host_syncs == 1 for s in got[3][1:])
    monkeypatch.delenv("NEUTRAL_STREAM_QUEUES")
    got = _run(iface, prob, cs, 2, 4)
    _same(want, got)
    stats = got[3]
    assert stats[0].stream_passes > 2, "the case no longer migrates"
    assert stats[0].host_syncs > 1          # first step: batches of 2, 2, 4, 8 ... passes
    for s in stats[1:]:
        assert s.host_syncs == 1, (s.host_syncs, s.stream_passes, s.stream_passes_enqueued)
        assert s.stream_passes <= s.stream_passes_enqueued

    prob = make_problem("csp", nx=100, nparticles=50000, iterations=4, dt=1.0e-6)
    got = _run(iface, prob, cs, 2, 4)
    assert [s.host_syncs for s in got[3][1:]] == [1, 1, 1]


def test_a_step_that_outruns_the_plan_is_finished(iface, make_problem, cs, monkeypatch):
    """Step 1 at a tiny dt needs one stream pass; step 2 (same store, ten times the
    dt through a second problem object) needs several more than were enqueued."""
    monkeypatch.setenv("NEUTRAL_WINDOW_MIN_PARTICLES", "32")
    short = make_problem("stream", nx=400, nparticles=30000, iterations=2, dt=1.0e-9)
    long_ = make_problem("stream", nx=400, nparticles=30000, iterations=2, dt=1.0e-7)

    def lengthen(sim, tt):
        if tt == 2:
            sim.p = long_

    want = _run(iface, short, cs, 0, 2, between=lengthen)
    got = _run(iface, short, cs, 2, 2, between=lengthen)
    _same(want, got)
    s1, s2 = got[3]
    assert s1.stream_passes == 1
    # two passes were enqueued on the strength of step 1; the rest came in batches
    assert s2.stream_passes > 2 and s2.host_syncs > 1
    assert s2.stream_passes <= s2.stream_passes_enqueued


def test_the_soa_arrays_are_current_after_every_step(iface, make_problem, cs):
    """Default (eager) mode: the kernels that end a history write it to the
    interface's arrays; nothing is pending when solve_transport_2d returns."""
    prob = make_problem("csp", nx=100, nparticles=50000, iterations=3, dt=1.0e-6)
    ref = iface.Simulation(prob, *cs, variant=0)
    sim = iface.Simulation(prob, *cs, variant=2)
    ref.inject()
    sim.inject()
    for tt in (1, 2, 3):
        ref.step(tt)
        sim.step(tt)
        a, b = ref.particles.contents, sim.particles.contents
        for f in iface.F64_FIELDS:
            assert np.array_equal(iface.to_host(getattr(a, f), ref.n, np.float64),
                                  iface.to_host(getattr(b, f), sim.n, np.float64)), (tt, f)
        for f in iface.I32_FIELDS:
            assert np.array_equal(iface.to_host(getattr(a, f), ref.n, np.int32),
                                  iface.to_host(getattr(b, f), sim.n, np.int32)), (tt, f)
    ref.close()
    sim.close()


@pytest.mark.parametrize("deck,nx,n,dt,steps", [
    ("csp", 100, 50000, 1.0e-6, 4),
    ("scatter", 64, 30000, 2.0e-7, 3),    # every history goes through the collision stage
    ("stream", 200, 30000, None, 2),      # none does
])
def test_arrays_are_current_through_deaths_and_table_changes(iface, make_problem, cs, deck, nx, n,
                                                             dt, steps):
    """The write-back pass leaves particles alone that were dead when the step began (their
    arrays are final) and does nothing on an attempt that the table check abandons: the
    arrays are current after every step all the same -- for particles that died steps ago,
    and when a cs table changes in between (the step's kernels then run twice)."""
    kw = dict(nx=nx, nparticles=n, iterations=steps)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    ref = iface.Simulation(prob, *cs, variant=0)
    sim = iface.Simulation(prob, *cs, variant=2)
    ref.inject()
    sim.inject()
    for tt in range(1, steps + 1):
        if tt == 3:
            for s in (ref, sim):
                s._av.mul_(0.5)    # no longer identical tables: the cached view is stale
        ref.step(tt)
        sim.step(tt)
        a, b = ref.particle_arrays(), sim.particle_arrays()
        for f in a:
            assert np.array_equal(a[f], b[f]), (tt, f)
    assert a["dead"].sum() > 0 or deck == "stream"
    ref.close()
    sim.close()


def _run_alone(iface, prob, cs, variant, schedule, rounds=1):
    """One simulation stepped on its own (another store stepped in between would write
    this one's pending records back -- the record workspace is shared -- and hide what the
    schedule is there to expose).  schedule: (timestep, lazy or None, look) per step;
    returns the arrays at the steps that look, the per-step counters and the tally."""
    sim = iface.Simulation(prob, *cs, variant=variant)
    looks, counters = [], []
    try:
        for _ in range(rounds):
            sim.inject()
            for tt, lazy, look in schedule:
                if lazy is not None:
                    iface.set_lazy_export(lazy if variant == 2 else False)
                r = sim.step(tt)
                counters.append((r.nprocessed, r.facets, r.collisions))
                if look:
                    looks.append((tt, sim.particle_arrays()))
        tally = sim.tally_host()
    finally:
        iface.set_lazy_export(False)
        sim.close()
    return looks, counters, tally


def _same_looks(want, got):
    assert want[1] == got[1]
    for (tt, a), (_, b) in zip(want[0], got[0]):
        for f in a:
            assert np.array_equal(a[f], b[f]), (tt, f)
    assert np.linalg.norm(want[2] - got[2]) <= 1e-13 * np.linalg.norm(want[2])


@pytest.mark.parametrize("deck,nx,n,dt", [("csp", 100, 40000, 1.0e-6),
                                          ("split", 64, 30000, 2.0e-8)])  # deaths every step
def test_write_back_mode_can_change_between_steps(iface, make_problem, cs, deck, nx, n, dt):
    """Eager steps note where each particle's record is (slot_of_id); lazy steps do not, and
    whoever asks for the arrays then reads the ids out of the records first.  The dead are
    the exception: their record moves for the last time when it is carried over, in either
    mode.  Switching between the modes, and asking in between or not, always gives the
    arrays of variant 0."""
    prob = make_problem(deck, nx=nx, nparticles=n, iterations=6, dt=dt)
    schedule = ((1, True, False), (2, False, True), (3, True, True), (4, True, False),
                (5, False, True), (6, True, True))
    want = _run_alone(iface, prob, cs, 0, schedule)
    got = _run_alone(iface, prob, cs, 2, schedule)
    _same_looks(want, got)
    if deck == "split":
        # particles died during the lazy steps 3 and 4: their records were carried over for
        # good by lazy steps, and the eager step 5 had to find them
        alive = [c[0] for c in want[1]]
        assert alive[3] < alive[2] and alive[4] < alive[3], alive


@pytest.mark.parametrize("deck,nx,n,dt", [("csp", 128, 60000, 2.0e-6), ("stream", 200, 40000, None)])
def test_carried_start_survives_steps_that_do_not_keep_it(iface, make_problem, cs, deck, nx, n, dt):
    """The stream kernel starts histories from the cross section carried with each record's slot and
    from the first draw the counting sort makes (neutral_history.h: prologue_carried) -- except in the
    instantiations that look up and draw themselves (tile queues here).  A step of the second kind
    keeps nothing of the carried values, so the library marks them stale and the next step of the
    first kind looks them up afresh (refresh_micro_kernel) before it relies on them.  Tile queues
    switched on and off between the steps of ONE store: the arrays and the event counts of the
    over-particle kernel at every step, tallies to summation order."""
    kw = dict(nx=nx, nparticles=n, iterations=6)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    pattern = (False, True, False, False, True, False)   # tile queues per step

    def run(variant):
        sim = iface.Simulation(prob, *cs, variant=variant)
        sim.inject()
        out = []
        try:
            for tt, queues in enumerate(pattern, start=1):
                iface.set_stream_queues(queues and variant == 2)
                r = sim.step(tt)
                out.append(((r.nprocessed, r.facets, r.collisions, r.census), sim.particle_arrays(),
                            r.stats.stream_hops))
            tally = sim.tally_host()
        finally:
            iface.set_stream_queues(False)
            sim.close()
        return out, tally

    want, want_t = run(0)
    got, got_t = run(2)
    for tt, ((ev0, a0, _), (ev2, a2, hops)) in enumerate(zip(want, got), start=1):
        assert ev0 == ev2, tt
        for f in a0:
            assert np.array_equal(a0[f], a2[f]), (tt, f)
    assert np.linalg.norm(want_t - got_t) <= 1e-13 * np.linalg.norm(want_t)
    if deck == "stream":
        # (the steps with queues did hand histories on inside the launch: they were the other kind)
        assert got[1][2] > 0 and got[4][2] > 0 and got[0][2] == 0


@pytest.mark.parametrize("deck,dt", [("scatter", None), ("split", 2.0e-8)])
@pytest.mark.parametrize("lazy", [False, True])
def test_the_dead_keep_their_slots_and_cost_nothing(iface, make_problem, cs, lazy, deck, dt):
    """A particle that is dead when a step begins is carried over behind the live ones once,
    copied across to the other record buffer once, and from then on takes no part in the
    sort (the graveyard).  scatter: every particle dies in the first step, so later steps
    run with nothing but graveyard and an empty sort; split at a short dt: a few thousand
    die in every step.  The arrays still equal variant 0's at every look; a re-injection
    starts over."""
    kw = dict(nx=64, nparticles=20000, iterations=6)
    if dt is not None:
        kw["dt"] = dt
    prob = make_problem(deck, **kw)
    schedule = tuple((tt, lazy, tt in (1, 3, 4, 6)) for tt in range(1, 7))
    want = _run_alone(iface, prob, cs, 0, schedule, rounds=2)
    got = _run_alone(iface, prob, cs, 2, schedule, rounds=2)
    _same_looks(want, got)
    alive = [c[0] for c in want[1]][:6]
    assert alive[0] == 20000 and alive[-1] < alive[0] // 2, alive
    if deck == "split":
        assert all(x > y for x, y in zip(alive, alive[1:])), alive


def test_two_live_stores_under_lazy_export(iface, make_problem, cs):
    """The record workspace is shared: stepping a second, larger store must first
    write the first store's pending state back (round-1 advisor finding)."""
    small = make_problem("csp", nx=100, nparticles=20000, iterations=2, dt=1.0e-6)
    big = make_problem("csp", nx=128, nparticles=90000, iterations=2, dt=1.0e-6)
    want_small = _run(iface, small, cs, 0, 2)
    want_big = _run(iface, big, cs, 0, 1)
    iface.set_lazy_export(True)
    try:
        a = iface.Simulation(small, *cs, variant=2)
        b = iface.Simulation(big, *cs, variant=2)
        a.inject()
        b.inject()
        a.step(1)
        a.step(2)              # pending in the records
        b.step(1)              # larger store: the workspace is reallocated
        got_a, got_b = a.particle_arrays(), b.particle_arrays()
        for f in got_a:
            assert np.array_equal(got_a[f], want_small[0][f]), f
            assert np.array_equal(got_b[f], want_big[0][f]), f
        a.close()
        b.close()
    finally:
        iface.set_lazy_export(False)


@pytest.mark.parametrize("lazy", [False, True])
def test_writes_through_the_hooks_reach_the_next_step(iface, make_problem, cs, lazy):
    """A caller that rewrites particle arrays between steps (here: kills every third
    particle through neutral_hip_memcpy_h2d) is seen by the tiled variant, whose
    records would otherwise still hold the old state; neutral_hip_invalidate_particles
    does the same for writes the library cannot see (here: a torch kernel)."""
    import torch
    prob = make_problem("csp", nx=100, nparticles=30000, iterations=3, dt=1.0e-6)
    kill = np.zeros(30000, dtype=np.int32)
    kill[::3] = 1

    def through_hook(sim, tt):
        if tt == 2:
            iface.library().neutral_hip_memcpy_h2d(
                C.c_void_p(sim.particles.contents.dead), kill.ctypes.data, kill.nbytes)

    def behind_its_back(sim, tt):
        if tt == 2:
            # halve every weight with a copy the library does not see (the HIP
            # runtime's own hipMemcpy, device to device), and say so
            n = sim.n
            weights = iface.to_host(sim.particles.contents.weight, n, np.float64) * 0.5
            t = torch.from_numpy(weights).to(sim.device)
            torch.cuda.synchronize()
            iface.library().neutral_hip_invalidate_particles(sim.particles)
            hip = _loaded_hip_runtime()
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip.hipMemcpy(C.c_void_p(sim.particles.contents.weight),
                                 C.c_void_p(t.data_ptr()), n * 8, 3) == 0  # device to device

    iface.set_lazy_export(lazy)
    try:
        for between in (through_hook, behind_its_back):
            want = _run(iface, prob, cs, 0, 3, between=between)
            got = _run(iface, prob, cs, 2, 3, between=between)
            _same(want, got)
            if between is through_hook:
                assert got[2][1][0] <= 20000
    finally:
        iface.set_lazy_export(False)


def test_a_table_rewritten_in_place_is_noticed(iface, make_problem, cs):
    """The view of the tables (identical? bucketed index) is cached across steps and
    re-checked on the device: halving the absorb table in place between steps must
    switch the next step to the two-search path, with the right physics."""
    prob = make_problem("csp", nx=64, nparticles=8192, iterations=3, dt=2.0e-6)

    def rewrite(sim, tt):
        if tt == 2:
            sim._av.mul_(0.5)          # values only: same keys, no longer identical
        if tt == 3:
            sim._ak.mul_(1.0 + 2**-40)  # keys too: the bucketed index is stale
            sim._sk.mul_(1.0 + 2**-40)

    want = _run(iface, prob, cs, 0, 3, between=rewrite)
    got = _run(iface, prob, cs, 2, 3, between=rewrite)
    _same(want, got)
    assert [s.same_tables for s in got[3]] == [1, 0, 0]
    got1 = _run(iface, prob, cs, 1, 3, between=rewrite)
    _same(want, got1)
