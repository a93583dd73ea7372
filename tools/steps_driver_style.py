"""Per-step stage times of the driver-style run (bench.py --steps 20 --warmup 5: five warm-up steps,
re-injection, timesteps 1..20), default ABI: where the 50 ms per step go, step by step."""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from neutral_amd import cs_table, decks, host
from neutral_amd import interface as iface
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
iface.set_quiet(True); iface.set_lazy_export(False)
keys, values = cs_table.load()
with tempfile.TemporaryDirectory() as tmp:
    path = decks.write_deck("csp", os.path.join(tmp, "d.params"), nx=400, ny=400, nparticles=n, iterations=20)
    prob = host.setup_problem(path)
    sim = iface.Simulation(prob, keys, values, variant=2)
    sim.inject()
    for tt in range(1, 6):
        sim.step(tt)
    sim.inject(); sim.zero_tally(); torch.cuda.synchronize()
    tot = 0.0
    for tt in range(1, 21):
        t0 = time.perf_counter(); r = sim.step(tt); wall = 1e3 * (time.perf_counter() - t0); s = r.stats
        tot += wall
        print(f"step {tt:2d}: wall {wall:6.2f} ms | sort {s.sort_ms:5.2f} stream {s.stream_ms:6.2f} ({s.stream_passes} passes, {s.stream_clock_ghz:.2f} GHz) "
              f"collide {s.collide_ms:6.2f} ({s.collide_clock_ghz:.2f} GHz) export-tail {s.export_ms:5.2f} | live {r.nprocessed} suspended {s.suspended} "
              f"facets {r.facets:.3e} collisions {r.collisions:.3e} steals {s.steals} weighted {s.weighted_waves}")
    print(f"mean wall {tot / 20:.2f} ms")
