#!/usr/bin/env python3
"""Times solve_transport_2d steps of one workload for the library selected by
NEUTRAL_HIP_LIB (kernel experiments; not part of the product or the tests).

  NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_x.so python tools/ablate.py csp 400 10000000 10
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neutral_amd import cs_table, decks, host  # noqa: E402
from neutral_amd import interface as iface  # noqa: E402


def main():
    deck, nx, n, its = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    variant = int(sys.argv[5]) if len(sys.argv) > 5 else None
    iface.set_quiet(True)
    iface.set_lazy_export(os.environ.get("NEUTRAL_EAGER_EXPORT") != "1")
    keys, values = cs_table.load()
    with tempfile.TemporaryDirectory() as tmp:
        path = decks.write_deck(deck, os.path.join(tmp, "d.params"), nx=nx, ny=nx, nparticles=n,
                                iterations=its)
        prob = host.setup_problem(path)
        sim = iface.Simulation(prob, keys, values, variant=variant)
        sim.inject()
        sim.step(1)            # warm-up
        sim.inject()
        sim.zero_tally()
        tot_ms = 0.0
        tot_steps = 0
        per = []
        for tt in range(1, its + 1):
            r = sim.step(tt)
            tot_ms += r.kernel_ms
            tot_steps += r.particle_steps
            per.append(f"{r.kernel_ms:.1f}" + (f"/p{r.stats.stream_passes}" if r.stats.stream_passes > 1 else ""))
        tag = os.path.basename(os.environ.get("NEUTRAL_HIP_LIB", "default"))
        print(f"{tag:40s} {deck} nx={nx} n={n}: {tot_ms:9.1f} ms  {tot_steps / tot_ms / 1e6:8.3f} Gsteps/s"
              f"  tally={float(sim.tally.sum()):.6e}  per-step ms: {' '.join(per)}", flush=True)
        sim.close()


if __name__ == "__main__":
    main()
