"""Where the stream kernel's waves spend their cycles: runs one bench workload on a library built
with -DNEUTRAL_PHASE_CLOCK (make -C neutral_amd variant TAG=phase EXTRA=-DNEUTRAL_PHASE_CLOCK) and
prints the phase sums the kernel keeps (neutral_tiled.hip: PHASE).
  NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_phase.so python tools/micro/phase_clock.py csp 400 100000000 8"""
import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.getcwd())
from neutral_amd import cs_table, decks, host
from neutral_amd import interface as iface
iface.set_quiet(True); iface.set_lazy_export(True)
keys, values = cs_table.load()
deck, nx, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
names = ["taking work, barriers", "window flush / move", "refill (loads, prologue)", "stream pass outside the facet loops",
         "(unused)", "facet loop", "census / end + its stores", "hand-offs"]
lib = iface.library()
out = (C.c_ulonglong * 8)()
with tempfile.TemporaryDirectory() as tmp:
    path = decks.write_deck(deck, os.path.join(tmp, "d.params"), nx=nx, ny=nx, nparticles=n, iterations=steps)
    prob = host.setup_problem(path, decks.ARCH_WIDTH, decks.ARCH_HEIGHT)
    sim = iface.Simulation(prob, keys, values, variant=2)
    sim.inject()
    lib.neutral_hip_debug_phase_clock(out)
    tot_ms = 0.0
    facets = 0
    for tt in range(1, steps + 1):
        r = sim.step(tt)
        if tt > steps // 2:
            tot_ms += r.stats.stream_ms
            facets += r.facets
        else:
            lib.neutral_hip_debug_phase_clock(out)
    lib.neutral_hip_debug_phase_clock(out)
    total = float(sum(out))
    print(f"{deck} {nx} {n}: steps {steps // 2 + 1}..{steps}, stream stage {tot_ms:.2f} ms, facets {facets:.3e}")
    for k in range(8):
        print(f"  {names[k]:40s} {out[k]:.3e} cycles  {100.0 * out[k] / total:5.1f} %")
    sim.close()
