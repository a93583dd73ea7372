#!/usr/bin/env python3
"""One-off check at scale: the tiled pipeline (variant 2, default settings: the collision stage's
work stealing active at this size) against the over-particle kernel (variant 0) -- same event
counts, same particle bits, tallies equal up to summation order.
  python tools/micro/compare_variants.py csp 400 40000000 4"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neutral_amd import cs_table, decks, host  # noqa: E402
from neutral_amd import interface as iface  # noqa: E402


def run(prob, keys, values, variant, steps):
    sim = iface.Simulation(prob, keys, values, variant=variant)
    sim.inject()
    ev, steals = [], 0
    for tt in range(1, steps + 1):
        r = sim.step(tt)
        ev.append((r.nprocessed, r.facets, r.collisions, r.census))
        steals += r.stats.steals
    arrays = sim.particle_arrays()
    tally = sim.tally_host()
    sim.close()
    return ev, arrays, tally, steals


def main():
    deck, nx, n, steps = sys.argv[1], int(sys.argv[2]), int(float(sys.argv[3])), int(sys.argv[4])
    iface.set_quiet(True)
    iface.set_lazy_export(False)
    keys, values = cs_table.load()
    with tempfile.TemporaryDirectory() as tmp:
        path = decks.write_deck(deck, os.path.join(tmp, "d.params"), nx=nx, ny=nx, nparticles=n,
                                iterations=steps)
        prob = host.setup_problem(path)
        ev0, a0, t0, _ = run(prob, keys, values, 0, steps)
        ev2, a2, t2, steals = run(prob, keys, values, 2, steps)
    same_events = ev0 == ev2
    fields = {f: bool(np.array_equal(a0[f], a2[f])) for f in a0}
    rel = float(np.linalg.norm(t0 - t2) / np.linalg.norm(t0))
    print(f"{deck} {nx}^2 {n} particles {steps} steps: events equal {same_events}; steals {steals}; "
          f"particle fields bit-equal {all(fields.values())} ({sum(fields.values())}/{len(fields)}); "
          f"tally rel L2 {rel:.2e}")
    if not (same_events and all(fields.values()) and rel < 1e-12):
        print({f: ok for f, ok in fields.items() if not ok})
        sys.exit(1)


if __name__ == "__main__":
    main()
