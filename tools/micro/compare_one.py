import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from neutral_amd import cs_table, decks, host
from neutral_amd import interface as iface
iface.set_quiet(True); iface.set_lazy_export(False)
keys, values = cs_table.load()
deck, nx, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
with tempfile.TemporaryDirectory() as tmp:
    path = decks.write_deck(deck, os.path.join(tmp, "d.params"), nx=nx, ny=nx, nparticles=n, iterations=steps)
    prob = host.setup_problem(path, decks.ARCH_WIDTH, decks.ARCH_HEIGHT)
    out = {}
    for variant in (0, 2):
        sim = iface.Simulation(prob, keys, values, variant=variant)
        sim.inject()
        ev = []
        for tt in range(1, steps + 1):
            r = sim.step(tt)
            s = r.stats
            ev.append((r.nprocessed, r.facets, r.collisions, r.census))
            if s.aborted:
                print("ABORTED histories: stopping here", s.aborted, flush=True)
                sys.exit(5)
            print(f"variant {variant} step {tt}: {ev[-1]} ms {s.kernel_ms:.3f} stream {s.stream_ms:.3f} passes {s.stream_passes} hops {s.stream_hops} overflows {s.stream_overflows} aborted {s.aborted}", flush=True)
        out[variant] = (ev, sim.particle_arrays(), sim.tally_host())
        sim.close()
    same = out[0][0] == out[2][0]
    fields = {f: bool(np.array_equal(out[0][1][f], out[2][1][f])) for f in out[0][1]}
    rel = float(np.linalg.norm(out[0][2] - out[2][2]) / np.linalg.norm(out[0][2]))
    print("events equal", same, "fields", fields, "tally rel", rel)
