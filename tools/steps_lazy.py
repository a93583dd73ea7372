import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from neutral_amd import cs_table, decks, host
from neutral_amd import interface as iface
lazy = int(sys.argv[1]); n = 100000000
iface.set_quiet(True); iface.set_lazy_export(bool(lazy))
keys, values = cs_table.load()
with tempfile.TemporaryDirectory() as tmp:
    path = decks.write_deck("csp", os.path.join(tmp, "d.params"), nx=400, ny=400, nparticles=n, iterations=10)
    prob = host.setup_problem(path)
    sim = iface.Simulation(prob, keys, values, variant=2)
    sim.inject(); torch.cuda.synchronize()
    for tt in range(1, 9):
        r = sim.step(tt); s = r.stats
        print(f"lazy {lazy} pools {os.environ.get('NEUTRAL_CU_POOLS','1')} step {tt}: collide {s.collide_ms:6.2f} ms ({s.collide_clock_ghz:.2f} GHz) stream {s.stream_ms:6.2f} suspended {s.suspended} collisions {r.collisions:.3e} requeued {s.requeued} passes {s.collide_passes} refused {s.steals_refused}")
