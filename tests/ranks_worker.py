"""One rank of a multi-rank run on a shared GPU (started by tests/test_decomposition.py
with RANK / WORLD_SIZE / MASTER_PORT set): steps a deck with the mesh decomposed over
the ranks (or the particles sharded, mode "shard") and leaves what it holds in
<out>/rank<r>.npz -- particle ids and state, its block of the tally, its event counts."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from neutral_amd import cs_table, host  # noqa: E402
from neutral_amd import interface as iface  # noqa: E402


def main():
    deck, out, steps, px, py, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), sys.argv[6]
    flux = len(sys.argv) > 7 and sys.argv[7] == "flux"
    iface.set_quiet(True)
    iface.set_device(0)
    transport = iface.comm_start()
    rank = iface.library().neutral_hip_comm_rank()
    prob = host.setup_problem(deck)
    keys, values = cs_table.load()
    sim = iface.Simulation(prob, keys, values, variant=2,
                           domain=(px, py) if mode == "domain" else None, scalar_flux=flux)
    sim.inject()
    events, counts, syncs, collectives, summed_over = [], [sim.n], [], [], []
    for tt in range(1, steps + 1):
        r = sim.step(tt)
        events.append((r.nprocessed, r.facets, r.collisions, r.census))
        counts.append(sim.n)
        syncs.append(r.stats.host_syncs)
        collectives.append(r.stats.host_collectives)
        summed_over.append(r.stats.exchange_ranks)
    arrays = sim.particle_arrays()
    ids = sim.particle_keys() if mode == "domain" else \
        (np.arange(sim.n, dtype=np.uint32) + np.uint32(sim.pid_base))
    np.savez(os.path.join(out, f"rank{rank}.npz"), ids=ids, tally=sim.tally_host(),
             flux=sim.flux.cpu().numpy() if flux else np.zeros(0),
             block=np.array([sim.x_off, sim.y_off, sim.lnx, sim.lny]),
             events=np.array(events, dtype=np.int64), counts=np.array(counts),
             transport=np.array([transport]), **arrays)
    sim.validate()
    sim.close()
    iface.library().neutral_hip_comm_barrier()
    iface.library().neutral_hip_comm_stop()
    print(json.dumps({"rank": rank, "counts": counts, "syncs": syncs, "collectives": collectives,
                      "exchange_ranks": summed_over}))


if __name__ == "__main__":
    main()
