"""Per-step collision-stage figures of csp 400^2 at any particle count (default: the share one of eight ranks
holds of the bench workload): histories per wave, collisions per pass (of 64 lanes), Gcollisions/s."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from neutral_amd import cs_table, decks, host
from neutral_amd import interface as iface
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500000
iface.set_quiet(True); iface.set_lazy_export(True)
keys, values = cs_table.load()
with tempfile.TemporaryDirectory() as tmp:
    path = decks.write_deck("csp", os.path.join(tmp, "d.params"), nx=400, ny=400, nparticles=n, iterations=20)
    prob = host.setup_problem(path)
    sim = iface.Simulation(prob, keys, values, variant=2)
    sim.inject()
    for tt in range(1, 6):
        sim.step(tt)
    sim.inject(); sim.zero_tally(); torch.cuda.synchronize()
    for tt in range(1, 21):
        r = sim.step(tt); s = r.stats
        if s.collide_passes:
            print(f"step {tt:2d}: collide {s.collide_ms:6.2f} ms  suspended {s.suspended:8d} ({s.suspended / 4096:6.1f} per wave)  collisions {r.collisions:.3e} "
                  f"passes {s.collide_passes:9d}  lanes/pass {r.collisions / s.collide_passes:5.1f}  passes/wave {s.collide_passes / 4096:7.1f}  "
                  f"Gcoll/s {r.collisions / s.collide_ms / 1e6:5.1f}  us/pass/wave {1e3 * s.collide_ms / (s.collide_passes / 4096):6.2f}  requeued {s.requeued} steals {s.steals}")
