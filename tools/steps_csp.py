import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neutral_amd import cs_table, decks, host
from neutral_amd import interface as iface
n = int(sys.argv[1]); variant = int(sys.argv[2])
iface.set_quiet(True); iface.set_lazy_export(True)
keys, values = cs_table.load()
with tempfile.TemporaryDirectory() as tmp:
    path = decks.write_deck("csp", os.path.join(tmp, "d.params"), nx=400, ny=400, nparticles=n, iterations=10)
    prob = host.setup_problem(path)
    sim = iface.Simulation(prob, keys, values, variant=variant)
    sim.inject(); sim.step(1); sim.inject(); sim.zero_tally()
    for tt in range(1, 11):
        r = sim.step(tt); s = r.stats
        print(f"step {tt}: live {r.nprocessed} suspended {s.suspended} collisions {r.collisions} "
              f"coll/susp {r.collisions/max(1,s.suspended):.0f} stream {s.stream_ms:.2f} ms collide {s.collide_ms:.2f} ms "
              f"sort {s.sort_ms:.2f} k2facets {r.facets - s.stream_facets} requeued {s.requeued} lanes/pass {r.collisions/max(1,s.collide_passes):.1f} -> {r.collisions/max(1e-9,s.collide_ms)/1e6:.1f} Gcoll/s")
