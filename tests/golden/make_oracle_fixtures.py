#!/usr/bin/env python3
"""Writes tests/golden/oracle_<deck>.npz: per-particle end state, per-cell tally
and event counts of the CPU oracle on four small problems (1024 particles,
32x32 mesh, 2 timesteps, single OpenMP thread so the tally is reproducible to
the bit).

These vectors come from oracle/neutral_oracle.c, NOT from the reference (whose
omp3 backend cannot be built here): they freeze the pinned
oracle so that a later edit to it cannot go unnoticed, and they let the GPU
tests check the HIP path against committed numbers.

Run: python tests/golden/make_oracle_fixtures.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle_binding as ob  # noqa: E402
from neutral_amd import cs_table, decks, host  # noqa: E402

CASES = {
    # deck: (nx, nparticles, iterations, dt)
    "scatter": (32, 1024, 2, 1.0e-7),
    "stream": (32, 1024, 2, 1.0e-7),
    "csp": (32, 1024, 2, 2.0e-6),
    "split": (32, 1024, 2, 1.0e-6),
}


def main():
    keys, values = cs_table.load()
    ob.lib().orc_set_num_threads(1)
    with tempfile.TemporaryDirectory() as tmp:
        for deck, (nx, n, its, dt) in CASES.items():
            path = decks.write_deck(deck, os.path.join(tmp, deck + ".params"), nx=nx, ny=nx,
                                    nparticles=n, iterations=its, dt=dt)
            prob = host.setup_problem(path)
            run = ob.OracleRun(prob, keys, values)
            run.inject()
            events = []
            for tt in range(1, its + 1):
                r = run.step(tt)
                events.append((r.nprocessed, r.facets, r.collisions, r.census))
            out = {f: getattr(run.particles, f) for f in ob.F64_FIELDS + ob.I32_FIELDS}
            np.savez_compressed(os.path.join(HERE, f"oracle_{deck}.npz"), tally=run.tally,
                                events=np.array(events, dtype=np.int64),
                                config=np.array([nx, n, its], dtype=np.int64), dt=np.float64(dt),
                                **out)
            print(deck, events, float(run.tally.sum()))


if __name__ == "__main__":
    main()
