"""Pins the CPU oracle (oracle/neutral_oracle.c) to the reference.

* Threefry2x64-20: bit for bit against the reference's own Random123 header
  compiled in place (oracle/_ref) and against the vectors in SURVEY.md 8(c);
* cross-section lookup: against the survey's recorded values;
* the whole history loop: against the event counts / tallies of the reference's
  omp3 backend recorded in BASELINE.md section 2 (exact integers), and against
  problems/neutral.tests (reference's own known answers, 1e-3).
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
from neutral_amd import decks


def _h(x):
    return int(x, 16)


def test_threefry_known_answers(pins):
    for v in pins["threefry2x64_20"]:
        out = ob.threefry(_h(v["ctr"][0]), _h(v["ctr"][1]), _h(v["key"][0]), _h(v["key"][1]))
        assert out == (_h(v["out"][0]), _h(v["out"][1]))


@pytest.mark.skipif(not ob.have_ref_threefry(), reason="oracle/_ref not built")
def test_threefry_matches_reference_random123(pins):
    # the recorded vectors really are what the reference header produces ...
    for v in pins["threefry2x64_20"]:
        out = ob.ref_threefry(_h(v["ctr"][0]), _h(v["ctr"][1]), _h(v["key"][0]), _h(v["key"][1]))
        assert out == (_h(v["out"][0]), _h(v["out"][1]))
    # ... and the restatement agrees with it on 20 000 random and edge inputs
    rng = np.random.default_rng(20261003)
    words = rng.integers(0, 2**64, size=(20000, 4), dtype=np.uint64)
    edge = [0, 1, 2**32 - 1, 2**32, 2**63, 2**64 - 1]
    cases = [tuple(int(w) for w in row) for row in words]
    cases += [(a, 0, b, c) for a in edge for b in edge for c in edge]
    for c0, c1, k0, k1 in cases:
        assert ob.threefry(c0, c1, k0, k1) == ob.ref_threefry(c0, c1, k0, k1)


def test_random_numbers_unit_interval():
    # omp3/neutral.c:646-651: u64 * 2^-64 + 2^-65, so rn in (0, 1]
    r0, r1 = ob.generate_random_numbers(12345, 1, 7)
    a, b = ob.threefry(7, 0, 12345, 1)
    assert r0 == float(a) * 2.0**-64 + 2.0**-65
    assert r1 == float(b) * 2.0**-64 + 2.0**-65
    rs = [ob.generate_random_numbers(p, 3, c) for p in range(200) for c in range(5)]
    flat = np.array(rs).ravel()
    assert flat.min() > 0.0 and flat.max() <= 1.0
    assert 0.45 < flat.mean() < 0.55


def test_cs_lookup_known_answers(pins, cs):
    table = ob.CsTable(*cs)
    for e in pins["cs_lookup"]:
        value, index = table.lookup(e["energy"])
        assert index == e["index"]
        assert value == pytest.approx(e["value"], rel=1e-14)


def test_cs_lookup_brackets(cs):
    keys, values = cs
    table = ob.CsTable(keys, values)
    rng = np.random.default_rng(7)
    idx = rng.integers(0, len(keys) - 1, size=500)
    for i in idx:
        # exactly on a key: bracket starts there, value is the tabulated one
        v, ind = table.lookup(float(keys[i]))
        assert ind == i and v == values[i]
        # just below the next key: still the same bracket
        e = np.nextafter(keys[i + 1], 0.0)
        v, ind = table.lookup(float(e))
        assert ind == i
        lo, hi = sorted((values[i], values[i + 1]))
        assert lo <= v <= hi
    # matches numpy's searchsorted bracket + the interpolation formula
    es = np.exp(rng.uniform(np.log(keys[0] * 1.001), np.log(keys[-1] * 0.999), size=2000))
    for e in es:
        v, ind = table.lookup(float(e))
        j = int(np.searchsorted(keys, e, side="right") - 1)
        assert ind == j
        expect = values[j] + ((e - keys[j]) / (keys[j + 1] - keys[j])) * (values[j + 1] - values[j])
        assert v == expect


def _run_oracle(make_problem, cs, name, nx, nparticles, iterations):
    prob = make_problem(name, nx=nx, nparticles=nparticles, iterations=iterations)
    run = ob.OracleRun(prob, *cs)
    run.inject()
    facets = collisions = 0
    last = None
    for tt in range(1, prob.niters + 1):
        last = run.step(tt)
        facets += last.facets
        collisions += last.collisions
    return run, facets, collisions, last


@pytest.mark.parametrize("i", range(5))
def test_oracle_reproduces_recorded_omp3_runs(pins, make_problem, cs, i):
    r = pins["omp3_runs"][i]
    run, facets, collisions, last = _run_oracle(make_problem, cs, r["deck"], r["nx"],
                                                r["nparticles"], r["iterations"])
    if r.get("facets") is not None:
        assert facets == r["facets"]
    if r.get("collisions") is not None:
        assert collisions == r["collisions"]
    if "collisions_rounded" in r:
        assert collisions == pytest.approx(r["collisions_rounded"], rel=5e-4)
    if "last_step_processed" in r:
        assert last.nprocessed == r["last_step_processed"]
    assert run.tally_sum() == pytest.approx(r["tally"], rel=1e-13)


def _known_answer(make_problem, cs, name, pins=None):
    d = decks.STANDARD_DECKS[name]
    run, facets, collisions, _ = _run_oracle(make_problem, cs, name, d["nx"], d["nparticles"],
                                             d["iterations"])
    expected = decks.KNOWN_ANSWERS[name]
    assert abs(run.tally_sum() - expected) / expected < decks.VALIDATE_TOLERANCE
    if pins is not None:
        # the omp3 backend's own output at this size, recorded by the survey
        rec = next(r for r in pins["omp3_default_runs"] if r["deck"] == name)
        if rec["facets"] is not None:
            assert facets == rec["facets"]
        if rec["collisions"] is not None:
            assert collisions == rec["collisions"]
        assert run.tally_sum() == pytest.approx(rec["tally"], rel=1e-13)


def test_known_answer_stream_default_deck(make_problem, cs, pins):
    """problems/neutral.tests:2 at the deck's default size (4000^2, 1e6 particles)."""
    _known_answer(make_problem, cs, "stream", pins)


@pytest.mark.fullkat
@pytest.mark.skipif(os.environ.get("NEUTRAL_FULL_KATS") != "1",
                    reason="minutes of CPU; set NEUTRAL_FULL_KATS=1 (log: oracle/pins/)")
@pytest.mark.parametrize("name", ["csp", "scatter"])
def test_known_answer_default_decks(make_problem, cs, name, pins):
    """problems/neutral.tests:1,3 at the decks' default sizes."""
    _known_answer(make_problem, cs, name, pins)
