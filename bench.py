#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the over-particle transport path on MI355X.

Workload (BASELINE.json: the metric is quoted on "problems/csp at 400x400,
1e8 particles"): the csp deck at nx = ny = 400 with 1e8 source particles.  One
bench "step" is one timestep of solve_transport_2d over every particle (the
deck runs 10 of them).  With N GPUs there is one process per GPU; the library
shards the 1e8 particles into N contiguous id ranges (strong scaling), each rank
tallies privately and solve_transport_2d ends every step with one all-reduce of
the 400x400 f64 tally over RCCL (include/neutral_hip.h, "ranks").

Timed region: W untimed warm-up timesteps, then the particles are re-injected
and the tally zeroed, then exactly K timesteps (master_key 1..K) between barrier +
torch.cuda.synchronize() pairs; MAX over ranks.  Inputs are resident in HBM
before the timed region starts.  `value` is measured the way an unmodified
main.c drives the library (the SoA particle arrays are current after every
step); `lazy_export` is a second timed region with neutral_hip_set_lazy_export.

One JSON line on rank 0 (see README "bench contract"), with
  roofline     the bound that holds for the dominant kernel -- vector instruction
               issue: issue cycles per launch (wave-level VALU instructions by class
               from the committed PMC passes, profiles/pmc_per_event.json, scaled
               by THIS run's event counts, priced with the measured cycles per class)
               / mean HIP-event duration of the kernel in this run / SIMD cycles
               available; next to it the HBM view SURVEY.md 8(d) asks for:
               algorithmic bytes, the compulsory floor B_hbm_min and measured traffic
  cpu_baseline the CPU oracle timed on this host's cores on a bounded sample of
               the same workload (rank 0, N = 1 only)
  parity_vs_cpu  the HIP path re-run on that same sample: per-cell tally L2
               against the oracle's tally ("tally L2 vs omp3"), exact event counts
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# vector issue: 256 CUs x 4 SIMDs, 2.4 GHz (MI355X_MICROARCH.md)
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
# Cycles one wave64 instruction holds its SIMD's vector issue: measured PER OPCODE for every
# opcode of the two hot loops (tools/micro/valu_opcodes.hip -> profiles/valu_cycles.json, four
# waves per SIMD, v_mul_f64 = 4): f64 arithmetic, 64-bit integer adds, conversions, compares,
# v_alignbit_b32, v_cndmask_b32, three-operand and carry 32-bit ops 3.5-4.4; v_rcp/rsq_f64
# 13.5; plain two-operand 32-bit ops (v_xor_b32, v_add_u32, v_mov_b32 ...) 2.3-2.5.  The PMC
# instruction classes cannot tell a 2.4-cycle v_xor_b32 from a 3.6-cycle v_alignbit_b32, the
# listing can: tools/isa_histogram.py prices the hot loop of each kernel opcode by opcode
# (profiles/isa_mix.json: mean issue cycles per vector instruction of the collision pass and
# of the facet trip), and the roofline is
#     wave-level vector instructions of the launch (PMC, per event x this run's events)
#       x mean cycles per instruction of the kernel's hot loop
#       / (kernel time x 1024 SIMDs x 2.4 GHz).
# What is left of a bracket is the share of the loop's cycles priced by an opcode FAMILY
# rule instead of a measurement (a few compares): those at 2 cycles (`frac_low`) or at their
# family's cost (`frac`, `frac_high` with them at 4.4).

WORKLOADS = {
    # name: (deck, nx, nparticles, deck iterations)
    "csp": ("csp", 400, 100_000_000, 10),
    "stream": ("stream", 400, 10_000_000, 1),
    "scatter": ("scatter", 400, 100_000_000, 2),
    "split": ("split", 800, 100_000_000, 1),
    # the reference's decks as shipped (problems/*.params: 4000^2 cells)
    "stream4000": ("stream", 4000, 1_000_000, 1),
    "csp4000": ("csp", 4000, 1_000_000, 10),
    "scatter4000": ("scatter", 4000, 10_000_000, 2),
    "split4000": ("split", 4000, 1_000_000, 1),
}


def algorithmic_bytes(histories, facets, collisions, census, same_tables):
    """Global-memory bytes the algorithm must touch: per history
    152 B of particle state in+out, 8 B density and one pair of cs lookups; per
    event 32 B of edges; per collision one more pair of lookups; per facet 16 B
    tally RMW + 8 B density; per census 16 B tally RMW.  A pair of lookups is
    15 probes x 16 B + 16 B of values per table = 512 B, or 272 B when both
    tables are the same data and the search is shared."""
    pair = 272 if same_tables else 512
    events = facets + collisions + census
    return (histories * (152 + 8 + pair) + events * 32 + collisions * pair +
            facets * 24 + census * 16)


def hbm_floor_bytes(histories, nx, ny, table_entries):
    """B_hbm_min of SURVEY.md 8(d): what must cross HBM when everything but the
    particles is cached -- 152 B of particle state per history, the two tables
    (keys + values) and the mesh arrays (density, tally) once."""
    return histories * 152 + 2 * 2 * 8 * table_entries + 2 * 8 * nx * ny


def kernel_events(results):
    """What each kernel of the tiled pipeline handled over the timed launches of this
    rank (the library's per-step statistics)."""
    if results[0].stats.variant != 2:
        name = "history_kernel" if results[0].stats.variant == 0 else "history_regroup_kernel"
        return {name: {"histories": sum(r.nprocessed for r in results),
                       "facets": sum(r.facets for r in results),
                       "collisions": sum(r.collisions for r in results),
                       "census": sum(r.census for r in results),
                       "ms": sum(r.kernel_ms for r in results)}}
    ev = {"stream_kernel": {"histories": 0, "facets": 0, "collisions": 0, "census": 0, "ms": 0.0},
          "history_regroup_kernel": {"histories": 0, "facets": 0, "collisions": 0, "census": 0,
                                     "ms": 0.0}}
    for r in results:
        st = r.stats
        s, c = ev["stream_kernel"], ev["history_regroup_kernel"]
        s["histories"] += r.nprocessed
        s["facets"] += st.stream_facets
        s["census"] += st.stream_census
        s["ms"] += st.stream_ms
        c["histories"] += st.suspended
        c["facets"] += r.facets - st.stream_facets
        c["collisions"] += r.collisions
        c["census"] += r.census - st.stream_census
        c["ms"] += st.collide_ms
    return ev


# the event a kernel's instruction and traffic counts scale with
PRIMARY_EVENT = {"stream_kernel": "facets", "history_regroup_kernel": "collisions",
                 "history_kernel": "collisions"}


FLUX = False  # --flux: the committed PMC coefficients are those of the kernels without it


SHARE_OF = None  # N > 1 (or --nparticles = workload / N): prefer the coefficients of that share


def profile_entry(deck, nx, variant, kernel):
    """The committed rocprofv3 PMC figures of `kernel` for this deck and mesh
    (profiles/pmc_per_event.json, made by tools/pmc_events.py from separate --pmc
    passes of this same command): counters PER PRIMARY EVENT of the profiled launches,
    or None when none is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_per_event.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except OSError:
        return None
    found = None
    for e in table.get("entries", []):
        if (e["deck"], e["nx"], e["variant"], e["kernel"], bool(e.get("flux", False))) == \
                (deck, nx, variant, kernel, FLUX):
            if e.get("share_of") == SHARE_OF:
                return e
            if e.get("share_of") is None:
                found = e   # (no coefficients of that share: the whole workload's)
    return found


HOT_LOOP = {"stream_kernel": "facet", "history_regroup_kernel": "collide",
            "history_kernel": "collide"}


def isa_mix(kernel):
    """Mean issue cycles per vector instruction of the kernel's hot loop and how much of
    that is priced by measured opcodes (profiles/isa_mix.json, tools/isa_histogram.py), and
    whether the listing it was made from is that of the device sources in this tree."""
    try:
        with open(os.path.join(ROOT, "profiles", "isa_mix.json")) as f:
            table = json.load(f)
    except OSError:
        return None
    mix = table.get(HOT_LOOP.get(kernel, ""))
    if mix is None:
        return None
    mix = dict(mix)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_histogram
        mix["listing_matches_sources"] = table.get("source_sha16") == isa_histogram.source_sha16()
    except Exception:
        mix["listing_matches_sources"] = None
    return mix


# full-load and idle readings of the hardware-only figure below, on kernels whose load is known by
# construction (tools/micro/pmc_calibration.hip, profiles/r05/pmc_calibration.log)
HW_VALU_BUSY_CALIBRATION = {"independent v_fma_f64, 4 waves/SIMD": 0.938, "1 wave/SIMD": 0.743,
                            "fma blocks separated by s_sleep": 0.400}
NOMINAL_GHZ = 2.4


def issue_roofline(deck, nx, variant, kernel, ev, launches, collide_passes=None, clock_ghz=None):
    """Vector-issue roofline of one kernel: issue cycles of THIS run's launches (wave-level
    vector instructions per event from the committed PMC passes x this run's events x the
    mean cost of an instruction of the kernel's hot loop, priced opcode by opcode) over
    the SIMD cycles its measured duration offers."""
    e = profile_entry(deck, nx, variant, kernel)
    mix = isa_mix(kernel)
    n = ev[PRIMARY_EVENT[kernel]]
    if e is None or mix is None or n == 0 or ev["ms"] <= 0:
        return None
    c = e["per_event"]
    insts = c["SQ_INSTS_VALU"] * n
    mean, lo_mean, hi_mean = (mix["mean_cycles_per_valu"], mix["mean_cycles_per_valu_low"],
                              mix["mean_cycles_per_valu_high"])
    seconds = ev["ms"] * 1e-3
    avail = SIMD_CYCLES_PER_S * seconds
    out = {"bound": "valu_issue", "kernel": kernel,
           "achieved": insts * mean / seconds / 1e9, "peak": SIMD_CYCLES_PER_S / 1e9,
           "unit": "G SIMD issue cycles/s", "frac": insts * mean / avail,
           "frac_low": insts * lo_mean / avail, "frac_high": insts * hi_mean / avail,
           "kernel_ms_avg": ev["ms"] / launches,
           "wave_valu_insts_per_launch": insts / launches,
           "mean_issue_cycles_per_valu_inst": mean,
           "share_of_loop_cycles_priced_by_measured_opcodes": mix["cycles_priced_by_measurement"],
           "valu_insts_per_event": c["SQ_INSTS_VALU"], "event": PRIMARY_EVENT[kernel],
           "events_per_launch": n / launches,
           "frac_kind": "modelled: a measured instruction count (PMC, per event x this run's events) "
                        "priced with the STATIC mix of the hot loop; prologue, refill, hand-back and "
                        "rare paths are priced as if they had the loop's mix.  frac_from_pass_count "
                        "(collision stage) is the same from the kernel's own dynamic pass count",
           "isa_listing_matches_sources": mix.get("listing_matches_sources"),
           "pricing": "instruction count: rocprofv3 --pmc SQ_INSTS_VALU per event x this run's "
                      "events; cycles per instruction: the hot loop's opcodes (tools/isa_histogram.py "
                      "on the shipped ISA) x the issue cost measured per opcode "
                      "(tools/micro/valu_opcodes.hip, v_mul_f64 = 4); clock 2.4 GHz nominal",
           "profiled": e.get("source", "profiles/pmc_per_event.json")}
    if collide_passes and HOT_LOOP.get(kernel) == "collide":
        # the same figure from the kernel's own dynamic count of wave-level collision passes x what a
        # pass issues: MEASURED per pass in the profiled run (refills and hand-backs included), and
        # -- an upper bound, kept for continuity -- the static trip of the pass loop, which also
        # counts the instructions of the blocks a trip rarely enters
        out["collision_passes_per_launch"] = collide_passes / launches
        out["frac_from_pass_count_static_trip"] = collide_passes * mix["issue_cycles_per_trip"] / avail
        out["static_trip_valu_insts"] = mix.get("valu_instructions")
        per_pass = e.get("valu_insts_per_collision_pass")
        if per_pass:
            out["valu_insts_per_pass_measured"] = per_pass
            out["frac_from_pass_count"] = collide_passes * per_pass * mean / avail
        else:
            out["frac_from_pass_count"] = out["frac_from_pass_count_static_trip"]
    # bracket: the pricing's own (opcodes priced by family), widened to hold the pass-count route
    routes = [out["frac"]] + ([out["frac_from_pass_count"]] if "frac_from_pass_count" in out else [])
    out["frac_bracket"] = {"low": min([out["frac_low"]] + routes), "high": max([out["frac_high"]] + routes),
                           "half_width_rel": (max([out["frac_high"]] + routes) - min([out["frac_low"]] + routes)) /
                           (2.0 * out["frac"])}
    # the clock there WAS: measured in the kernel's own launches of this run (one wave per launch,
    # shader-clock ticks per 100-MHz tick); `frac` divides by the nominal 2.4 GHz
    if clock_ghz:
        out["shader_clock_ghz_measured"] = clock_ghz
        out["frac_at_measured_clock"] = out["frac"] * NOMINAL_GHZ / clock_ghz
    # hardware only, from the profiled run's SQ counters (no pricing, no clock): quads of vector-ALU
    # activity over the SQ-busy cycles of the launches x 1024 SIMDs.  SQ_ACTIVE_INST_VALU counts one
    # 4-cycle quad per v_fma_f64 (calibration) and rounds cheaper 32-bit opcodes UP to a quad, so
    # this reads high on integer-heavy code; the calibration readings say what full and idle look like
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_BUSY_CYCLES"):
        out["hw"] = {
            "valu_active_share_of_sq_busy_cycles": 4.0 * c["SQ_ACTIVE_INST_VALU"] /
            (1024.0 * c["SQ_BUSY_CYCLES"] / 32.0),
            "calibration": HW_VALU_BUSY_CALIBRATION,
            "cycles_per_valu_inst_upper": 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"],
            "source": "SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES (32 SQ instances), SQ_INSTS_VALU of " +
                      str(e.get("source", "profiles/pmc_per_event.json"))}
        if e.get("kernel_ms_per_launch_same_flags") and e.get("dispatches_profiled"):
            busy = c["SQ_BUSY_CYCLES"] * e["events_profiled"] / 32.0 / e["dispatches_profiled"]
            out["hw"]["shader_clock_ghz_under_profiler"] = busy / (e["kernel_ms_per_launch_same_flags"] * 1e6)
    if c.get("SQ_ACTIVE_INST_VALU"):
        out["lane_utilisation"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / \
            (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_INST_ANY") is not None:
        out["wait_inst_any_share_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    return out


def hbm_view(deck, nx, variant, kernel, ev, launches, same_tables):
    """The HBM side of one kernel: SURVEY 8(d) algorithmic bytes and, from the
    FETCH_SIZE / WRITE_SIZE passes, the traffic that actually crossed HBM."""
    b = algorithmic_bytes(ev["histories"], ev["facets"], ev["collisions"], ev["census"],
                          same_tables)
    seconds = ev["ms"] * 1e-3
    out = {"algorithmic_bytes_per_launch": b / launches,
           "achieved_gbs_by_algorithmic_bytes": b / seconds / 1e9 if seconds > 0 else None,
           "note": "algorithmic bytes price every cs lookup at the reference's 15 probes "
                   "(SURVEY 8d); the bucketed index in LDS removes most of them and the rest "
                   "is served by L1/L2, so this figure can exceed the HBM peak: it is not a "
                   "roofline fraction"}
    e = profile_entry(deck, nx, variant, kernel)
    n = ev[PRIMARY_EVENT[kernel]]
    if e is not None and n and "hbm_bytes" in e["per_event"]:
        traffic = e["per_event"]["hbm_bytes"] * n
        out["traffic_bytes_per_launch"] = traffic / launches
        out["traffic_gbs"] = traffic / seconds / 1e9
        out["traffic_frac_of_hbm_peak"] = traffic / seconds / 1e9 / HBM_PEAK_GBS
    if e is not None and n and e["per_event"].get("TCC_REQ_sum"):
        c = e["per_event"]
        hits, misses = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        if hits + misses > 0:
            out["l2_hit_rate"] = hits / (hits + misses)
        out["l2_requests_per_event"] = c["TCC_REQ_sum"]
        out["l2_atomics_per_event"] = c.get("TCC_ATOMIC_sum", 0.0)
    return out


def one_rank_path():
    # (NEUTRAL_ONE_RANK_RECORD: another file, for the tests)
    return os.environ.get("NEUTRAL_ONE_RANK_RECORD") or \
        os.path.join(ROOT, "profiles", "one_rank_tally.json")


def one_rank_record(deck, nx, nparticles, steps):
    """What ONE rank computes for this workload and step count (profiles/one_rank_tally.json:
    event totals and the global tally of `bench.py --gpus 1`, recorded on the GPU box): the
    N > 1 bench line is checked against it -- event counts exactly, the all-reduced tally to
    1e-12 (summation order) -- because a run over several GPUs has no CPU leg of its own."""
    try:
        with open(one_rank_path()) as f:
            table = json.load(f)
    except OSError:
        return None
    for e in table.get("entries", []):
        if (e["deck"], e["nx"], e["nparticles"], e["steps"]) == (deck, nx, nparticles, steps):
            return e
    return None


def measured_copy_bandwidth(device):
    """On-box HBM stream-copy rate (GB/s, read + write bytes) of a 1-GiB f64
    device-to-device copy: the measured denominator SURVEY.md 8(d) asks for next
    to the 8 TB/s spec figure."""
    import torch
    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device=device)
    b = torch.empty_like(a)
    a.fill_(1.0)
    b.copy_(a)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    del a, b
    return 2.0 * n * 8 * reps / dt / 1e9


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="csp", choices=sorted(WORKLOADS))
    ap.add_argument("--nparticles", type=int, default=None,
                    help="override the workload's total particle count")
    ap.add_argument("--nx", type=int, default=None)
    ap.add_argument("--variant", type=int, default=2,
                    help="0 over-particle, 1 event-regrouped, 2 tiled (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lazy-leg", action="store_true",
                    help="skip the second timed region (lazy export)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "host"],
                    help="tally exchange for N > 1: RCCL over xGMI, or staged through the host")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses GPU 0 (needs --comm host)")
    ap.add_argument("--record-one-rank", action="store_true",
                    help="N = 1: write this run's event totals and global tally to "
                         "profiles/one_rank_tally.json (what N > 1 lines are checked against)")
    ap.add_argument("--flux", action="store_true",
                    help="keep the scalar-flux tally next to the energy deposition (SURVEY 8f-3: two "
                         "88-cell LDS windows instead of one of 128 cells)")
    ap.add_argument("--decompose", default=None, metavar="PXxPY",
                    help="N > 1: spatial domain decomposition (SURVEY 8f-4) -- every rank owns a block "
                         "of the mesh and the particles inside it; histories migrate between ranks")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target CPU work of the cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(deck, nx, its, target_seconds, tmp):
    """Times the CPU oracle (tests/oracle_binding.py) on all host cores over a
    bounded sample of the same workload: same deck, mesh and timestep count,
    fewer particles (steps/s is intensive in N: BASELINE.md section 2)."""
    import oracle_binding as ob
    from neutral_amd import cs_table, decks, host

    keys, values = cs_table.load()
    cores = ob.lib().orc_num_threads()
    quota = cpu_quota_cores()

    def run(n):
        path = decks.write_deck(deck, os.path.join(tmp, f"cpu_{n}.params"), nx=nx, ny=nx,
                                nparticles=n, iterations=its)
        prob = host.setup_problem(path, decks.ARCH_WIDTH, decks.ARCH_HEIGHT)
        r = ob.OracleRun(prob, keys, values)
        r.inject()
        steps = 0
        t0 = time.perf_counter()
        r.events = []
        for tt in range(1, its + 1):
            res = r.step(tt)
            steps += res.particle_steps
            r.events.append((res.nprocessed, res.facets, res.collisions, res.census))
        return steps, time.perf_counter() - t0, r

    # The static partition and the tally atomics of the omp3 scheme do not always
    # scale to every hardware thread, and a job's CPU quota may be far below the threads
    # it can see: probe thread counts from all of them down to the quota (and below) on a
    # small sample and time the real sample with the fastest, so the baseline is the CPU at
    # its best.  Threads are pinned (OMP_PROC_BIND / OMP_PLACES, set in main() before any
    # OpenMP runtime loads) and the line says so.
    n0 = 200_000
    best = None
    candidates = {cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8), max(1, cores // 16)}
    if quota:
        candidates |= {max(1, int(quota)), max(1, 2 * int(quota))}
    probed = []
    for threads in sorted((c for c in candidates if c <= cores), reverse=True):
        ob.lib().orc_set_num_threads(threads)
        s_, t_, _ = run(n0)
        probed.append({"threads": threads, "particle_steps_per_s": s_ / t_})
        if best is None or s_ / t_ > best[0]:
            best = (s_ / t_, threads, t_)
    cores = best[1]
    t0 = best[2]
    ob.lib().orc_set_num_threads(cores)
    n = int(min(max(n0, n0 * target_seconds / max(t0, 1e-3)), 20_000_000))
    n = max(n0, (n // 1000) * 1000)
    steps, secs, oracle_run = run(n)
    out = {"value": steps / secs, "unit": "particle-steps/s", "cores": cores, "kind": "port",
           "sample": f"{deck} {nx}x{nx}, {n} particles, {its} timesteps, "
                     f"{steps} particle-steps in {secs:.2f} s (CPU oracle, OpenMP static)",
           "proc_bind": os.environ.get("OMP_PROC_BIND"), "places": os.environ.get("OMP_PLACES"),
           "threads_visible": os.cpu_count(),
           "cpu_quota_cores": quota,
           "probed": probed,
           "cpu_model": cpu_model()}
    return out, oracle_run, n


def cpu_quota_cores():
    """CPU time this job may use, in cores (a container's cpu.max), or None when unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def main():
    # ROCr reads this at start-up: multi-process GPU work on this pool needs dmabuf IPC
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse_args()
    if args.gpus == 1 and not args.no_cpu_baseline:
        # the CPU baseline's threads stay where they are put (read by the OpenMP runtime when it
        # loads, i.e. before torch or the oracle pull one in)
        os.environ.setdefault("OMP_PROC_BIND", "close")
        os.environ.setdefault("OMP_PLACES", "cores")
    global FLUX
    FLUX = bool(args.flux)
    domain = None
    if args.decompose:
        px, _, py = args.decompose.lower().partition("x")
        domain = (int(px), int(py))
        if domain[0] * domain[1] != args.gpus:
            raise SystemExit(f"--decompose {args.decompose} needs --gpus {domain[0] * domain[1]}")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not under a launcher: start one (child process; nothing here has touched the GPU)
        port = os.environ.get("MASTER_PORT", "29517")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_device:
        if args.comm != "host":
            raise SystemExit("--share-device needs --comm host (RCCL wants one GPU per rank)")
        local_rank = 0
    if args.comm == "host":
        os.environ["NEUTRAL_HIP_COMM"] = "host"
    os.environ.setdefault("NEUTRAL_HIP_QUIET", "1")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: GPU {local_rank} is not visible "
                         f"({torch.cuda.device_count()} devices)")
    torch.cuda.set_device(local_rank)

    from neutral_amd import cs_table, decks, host
    from neutral_amd import interface as iface

    iface.set_device(local_rank)
    if world > 1 and "NEUTRAL_COMM_PORT" not in os.environ:
        # the library's ranks meet on a TCP port of their own; rank 0 picks a free one and
        # tells the others through the launcher's store at MASTER_ADDR:MASTER_PORT
        # (PyTorch as plumbing only: no process group, no collective)
        import socket
        from datetime import timedelta
        from torch.distributed import TCPStore
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"),
                         int(os.environ.get("MASTER_PORT", "29500")), world, rank == 0,
                         timeout=timedelta(seconds=120))
        if rank == 0:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                store.set("neutral_comm_port", str(sock.getsockname()[1]))
            # the word this launch's ranks greet rank 0 with (host/comms_ranks.c)
            store.set("neutral_comm_nonce", str(int.from_bytes(os.urandom(8), "little")))
        os.environ["NEUTRAL_COMM_PORT"] = store.get("neutral_comm_port").decode()
        os.environ.setdefault("NEUTRAL_COMM_NONCE", store.get("neutral_comm_nonce").decode())
    # the rank layer of the library: TCP rendezvous of the ranks, then RCCL on this rank's
    # GPU with a time limit; a rank that cannot get RCCL up makes every rank stage the
    # exchange through the host instead (reported below)
    transport = iface.comm_start() if world > 1 else iface.COMM_NONE
    lib = iface.library()

    iface.set_quiet(True)
    deck, nx, ntotal, deck_its = WORKLOADS[args.workload]
    nx = args.nx or nx
    ntotal = args.nparticles or ntotal
    K, W = args.steps, args.warmup
    keys, values = cs_table.load()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            lib.neutral_hip_comm_barrier()
        torch.cuda.synchronize()

    with tempfile.TemporaryDirectory() as tmp:
        path = decks.write_deck(deck, os.path.join(tmp, f"{deck}_r{rank}.params"), nx=nx,
                                ny=nx, nparticles=ntotal, iterations=max(K, 1))
        prob = host.setup_problem(path, decks.ARCH_WIDTH, decks.ARCH_HEIGHT)
        # shard=None: with several ranks inject_particles cuts this rank's share itself
        sim = iface.Simulation(prob, keys, values, device=local_rank, variant=args.variant,
                               scalar_flux=args.flux, domain=domain)

        def timed_region(lazy):
            """W warm-up steps, re-injection, K timed steps; returns (results, seconds,
            global event totals, write-back ms)."""
            iface.set_lazy_export(lazy)
            sim.inject()
            torch.cuda.synchronize()
            for tt in range(1, W + 1):
                t_w = time.perf_counter()
                sim.step(tt)
                step_wall_ms.setdefault("warmup", []).append(1e3 * (time.perf_counter() - t_w))
            sim.inject()
            sim.tally.zero_()
            if sim.flux is not None:
                sim.flux.zero_()
            fence()
            t0 = time.perf_counter()
            results = []
            walls = []
            for tt in range(1, K + 1):
                t_s = time.perf_counter()
                results.append(sim.step(tt))   # (synchronous on return: main.c stops its timer there)
                walls.append(1e3 * (time.perf_counter() - t_s))
            fence()
            elapsed = time.perf_counter() - t0
            step_wall_ms.setdefault("lazy" if lazy else "timed", walls)
            t_wb = time.perf_counter()
            lib.neutral_hip_sync_particles(sim.particles)   # lazy: the deferred write-back
            torch.cuda.synchronize()
            writeback_ms = 1e3 * (time.perf_counter() - t_wb)
            if world > 1:
                elapsed = lib.neutral_hip_comm_max(elapsed)
            # (with several ranks the library's event counts are already global sums)
            tot = {"facets": sum(r.facets for r in results),
                   "collisions": sum(r.collisions for r in results),
                   "census": sum(r.census for r in results),
                   "histories": sum(r.nprocessed for r in results)}
            return results, elapsed, tot, writeback_ms

        # host wall time of every solve_transport_2d call of this rank, by region (the first call of
        # the process carries what is set up lazily: RCCL's channels at the first exchange, the
        # table view, the workspace)
        step_wall_ms = {}
        # ---- the headline: the library as an unmodified main.c drives it ----
        results, elapsed, tot, _ = timed_region(lazy=False)
        stats = iface.last_step()
        # ---- what a first multi-GPU record needs to be read: every rank's own view ----
        ranks_view = None
        if world > 1:
            import ctypes as C
            first_call = (step_wall_ms.get("warmup") or step_wall_ms["timed"])[0]
            timed_walls = step_wall_ms["timed"]
            steady = timed_walls[1:] or timed_walls
            mine = [float(transport),
                    sum(r.stats.kernel_ms + r.stats.export_ms for r in results) / K,
                    sum(r.stats.stream_ms for r in results) / K,
                    sum(r.stats.collide_ms for r in results) / K,
                    sum(r.stats.exchange_ms for r in results) / K,
                    float(results[-1].stats.local_nprocessed),
                    float(sim.n),
                    first_call, timed_walls[0], sum(steady) / len(steady),
                    max(r.stats.exchange_ms for r in results)]
            t = torch.zeros(world * len(mine), dtype=torch.float64, device=sim.device)
            t[rank * len(mine):(rank + 1) * len(mine)] = torch.tensor(mine, dtype=torch.float64)
            torch.cuda.synchronize()
            lib.neutral_hip_comm_allreduce_f64(C.c_void_p(t.data_ptr()), t.numel(),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            rows = t.cpu().reshape(world, len(mine)).tolist()

            def spread(col):
                v = [row[col] for row in rows]
                return {"min": min(v), "max": max(v), "per_rank": v}
            names = {iface.COMM_NONE: "none", iface.COMM_RCCL: "rccl", iface.COMM_HOST: "host"}
            ranks_view = {
                "rccl_version": int(lib.neutral_hip_comm_rccl_version()),
                "transport_asked": args.comm,
                "transport_per_rank": [names.get(int(row[0]), "?") for row in rows],
                "device_ms_per_step": spread(1),   # kernels + write-back, HIP events, this rank
                "stream_ms_per_step": spread(2),
                "collide_ms_per_step": spread(3),
                "exchange_ms_per_step": spread(4),  # on the library's own stream
                "particles_alive_last_step": spread(5),
                "particles_in_shard": spread(6),
                # host wall time of solve_transport_2d, per rank: the process's FIRST call (RCCL
                # sets its channels up lazily at the first exchange; table view, workspace), the
                # first timed step, and the mean of the timed steps after it
                "first_call_wall_ms": spread(7),
                "first_timed_step_wall_ms": spread(8),
                "steady_step_wall_ms": spread(9),
            }
            # the exchange is 1.28 MB + 160 B per step: beyond half a millisecond on any rank it
            # is not latency-bound any more (a fallback transport, a slow link, a rank that
            # arrives late), and the line says so instead of letting the scaling figure carry it
            worst = max(row[10] for row in rows)
            bar = 0.5
            ranks_view["exchange_check"] = {
                "worst_step_ms_any_rank": worst, "mean_ms_per_step_max_over_ranks": spread(4)["max"],
                "bar_ms": bar, "ok": bool(spread(4)["max"] < bar),
                "note": "exchange_ms is HIP-event time on the library's own stream: pack, two "
                        "all-reduces (tally, step words), the add into the caller's mesh"}
            if not ranks_view["exchange_check"]["ok"] and rank == 0:
                print(f"bench.py: the tally exchange takes {spread(4)['max']:.3f} ms per step on the "
                      f"slowest rank (bar {bar} ms; worst single step {worst:.3f} ms): the N > 1 figure "
                      f"is bound by it, see ranks.exchange_ms_per_step", file=sys.stderr, flush=True)
            if domain is not None:
                mine2 = [sum(r.stats.exchange_rounds for r in results) / K,
                         sum(r.stats.emigrants for r in results) / K,
                         sum(r.stats.stream_passes for r in results) / K]
                t2 = torch.zeros(world * 3, dtype=torch.float64, device=sim.device)
                t2[rank * 3:(rank + 1) * 3] = torch.tensor(mine2, dtype=torch.float64)
                torch.cuda.synchronize()
                lib.neutral_hip_comm_allreduce_f64(C.c_void_p(t2.data_ptr()), t2.numel(),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                rows2 = t2.cpu().reshape(world, 3).tolist()
                ranks_view["decomposition"] = {
                    "grid": args.decompose,
                    "exchange_rounds_per_step": max(r2[0] for r2 in rows2),
                    "emigrants_per_step_per_rank": [r2[1] for r2 in rows2],
                    "emigrants_per_round": (sum(r2[1] for r2 in rows2) /
                                            max(1.0, max(r2[0] for r2 in rows2))),
                    "stream_passes_per_step_per_rank": [r2[2] for r2 in rows2]}
        particle_steps = tot["facets"] + tot["collisions"] + tot["census"]
        global_tally = float(sim.tally.sum().item())
        if domain is not None:
            # (every rank tallies the cells of its block: the global tally is the sum of the blocks')
            import ctypes as C
            g1 = torch.tensor([global_tally], dtype=torch.float64, device=sim.device)
            torch.cuda.synchronize()
            lib.neutral_hip_comm_allreduce_f64(C.c_void_p(g1.data_ptr()), 1,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            global_tally = float(g1.item())
        lazy = None
        l_results = None
        if not args.no_lazy_leg:
            l_results, l_elapsed, l_tot, l_wb = timed_region(lazy=True)
            iface.set_lazy_export(False)
            lazy = {"value": (l_tot["facets"] + l_tot["collisions"] + l_tot["census"]) / l_elapsed,
                    "ms_per_step": 1e3 * l_elapsed / K, "ms_per_writeback_on_demand": l_wb,
                    "mode": "neutral_hip_set_lazy_export(1): the SoA particle arrays are written "
                            "back on demand (nothing reads them between timesteps; main.c with "
                            "visit_dump = 0 does not either), not by every step",
                    "events_equal": l_tot == tot}

        if rank == 0:
            # rooflines from this rank's launches (HIP events recorded inside the C-ABI on
            # the stream the kernels run on); with several ranks the per-kernel event
            # counts are this rank's share only for the tiled pipeline's split -- the
            # library reports global sums there too, so scale by rank count
            kev = kernel_events(results)
            for ev in kev.values():
                for k in ("histories", "facets", "collisions", "census"):
                    ev[k] = ev[k] / world
            same = bool(stats.same_tables)
            variant = int(stats.variant)
            kernels = []

            def clock_of(name, res):
                """mean shader clock the kernel's launches of `res` ran at (this rank's, measured
                in the kernels), or None when the variant does not measure it"""
                field = {"stream_kernel": "stream_clock_ghz",
                         "history_regroup_kernel": "collide_clock_ghz"}.get(name)
                v = [getattr(r.stats, field) for r in res if field and getattr(r.stats, field) > 0]
                return (sum(v) / len(v)) if v else None
            global SHARE_OF
            full = WORKLOADS[args.workload][2]
            SHARE_OF = world if world > 1 else (round(full / ntotal) if ntotal < full and
                                                abs(full / ntotal - round(full / ntotal)) < 1e-9 else None)
            for name, ev in kev.items():
                kernels.append({"name": name, "ms_per_launch": ev["ms"] / K,
                                "events_per_launch": {k: ev[k] / K for k in
                                                      ("histories", "facets", "collisions",
                                                       "census")},
                                "valu_issue": issue_roofline(
                                    deck, nx, variant, name, ev, K,
                                    sum(r.stats.collide_passes for r in results) / world,
                                    clock_of(name, results)),
                                "hbm": hbm_view(deck, nx, variant, name, ev, K, same)})
            if variant == 2:
                kernels.append({"name": "tile sort (count, scan, place, chunks) + collision queue",
                                "ms_per_launch": sum(r.stats.sort_ms for r in results) / K})
            dom = max((k for k in kernels if "valu_issue" in k), key=lambda k: k["ms_per_launch"])
            histories_rank = tot["histories"] / world
            floor = hbm_floor_bytes(histories_rank / K, nx, nx, len(keys))
            traffic = sum((k.get("hbm") or {}).get("traffic_bytes_per_launch", 0.0)
                          for k in kernels if "hbm" in k)
            roofline = dict(dom["valu_issue"] or {"bound": "valu_issue", "kernel": dom["name"],
                                                  "achieved": None, "peak":
                                                  SIMD_CYCLES_PER_S / 1e9, "frac": None,
                                                  "unit": "G SIMD issue cycles/s",
                                                  "kernel_ms_avg": dom["ms_per_launch"],
                                                  "note": "no PMC coefficients committed for this "
                                                          "deck / mesh (profiles/pmc_per_event.json)"})
            if l_results is not None and roofline.get("frac") is not None:
                # the same kernel with the chip to itself: the lazy leg of this run has no write-back
                # pass, while the headline's runs BESIDE the collision stage on a second stream (the
                # library's split write-back) and is part of what that stage's duration measures
                lev = kernel_events(l_results).get(dom["name"])
                if lev:
                    for k in ("histories", "facets", "collisions", "census"):
                        lev[k] = lev[k] / world
                    alone = issue_roofline(deck, nx, variant, dom["name"], lev, K,
                                           sum(r.stats.collide_passes for r in l_results) / world,
                                           clock_of(dom["name"], l_results))
                    if alone:
                        roofline["alone"] = {
                            "frac": alone["frac"], "frac_low": alone["frac_low"], "frac_high": alone["frac_high"],
                            "frac_from_pass_count": alone.get("frac_from_pass_count"),
                            "frac_bracket": alone.get("frac_bracket"),
                            "shader_clock_ghz_measured": alone.get("shader_clock_ghz_measured"),
                            "frac_at_measured_clock": alone.get("frac_at_measured_clock"),
                            "kernel_ms_avg": alone["kernel_ms_avg"],
                            "note": "same kernel, same run, second timed region (neutral_hip_set_lazy_export(1): no "
                                    "write-back pass runs beside it); in the headline region the write-back of the "
                                    "histories that never collide runs on a second stream beside this kernel, and "
                                    "`frac` / `kernel_ms_avg` above include what that costs it"}
            roofline["traffic"] = (dom.get("hbm") or {}).get("traffic_bytes_per_launch")
            roofline["hbm"] = {
                "dominant_kernel": dom.get("hbm"),
                "b_hbm_min_per_step": floor,
                "traffic_per_step": traffic or None,
                "traffic_over_b_hbm_min": (traffic / floor) if traffic else None,
                "step_traffic_gbs": (traffic / (elapsed / K) / 1e9) if traffic else None,
                "peak_gbs": HBM_PEAK_GBS,
                "hbm_copy_measured_gbs": measured_copy_bandwidth(sim.device)}
            out = {
                "metric": "particle-steps/sec",
                "value": particle_steps / elapsed,
                "unit": "particle-steps/s",
                "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": 1e3 * elapsed / K,
                "higher_is_better": True,
                "scaling": "strong",
                "vs_baseline": None,
                "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": f"problems/{deck}.params at nx=ny={nx}, {ntotal} particles, "
                                       f"{K} timesteps" +
                                       (" (BASELINE.json: the metric is quoted on csp 400x400, 1e8 "
                                        "particles)" if args.workload == "csp" else
                                        f" (--workload {args.workload})"),
                           "deck": deck, "nx": nx, "ny": nx, "nparticles": ntotal,
                           "timesteps": K,
                           "parallelism": (f"mesh blocks {args.decompose}" if domain is not None
                                           else f"particle-shard x{world}"),
                           "kernel_variant": variant,
                           "particle_arrays": "current after every step (default ABI)",
                           "scalar_flux_tally": bool(args.flux),
                           "decomposition": args.decompose,
                           "tally_exchange": {iface.COMM_NONE: "none (one rank)",
                                              iface.COMM_RCCL: "RCCL all-reduce per step",
                                              iface.COMM_HOST: "staged through the host (TCP)"}
                           [transport]},
                "ns_per_particle_step": 1e9 * elapsed / particle_steps,
                "histories_per_s": tot["histories"] / elapsed,
                "events": tot,
                "global_tally": global_tally,
                "host_waits_per_step": int(stats.host_syncs),
                "stream_passes_per_step": int(stats.stream_passes),
                "stream_queue": {"hops_per_step": sum(r.stats.stream_hops for r in results) / K,
                                 "batches_per_step": sum(r.stats.stream_batches for r in results) / K,
                                 "overflows_per_step": sum(r.stats.stream_overflows for r in results) / K,
                                 "idle_polls_per_step": sum(r.stats.stream_idle_polls for r in results) / K},
                "tile_cells": int(stats.tile_cells),
                "lazy_export": lazy,
                "roofline": roofline,
                "kernels": kernels,
            }
            out["exchange"] = {"ranks_summed_over": int(stats.exchange_ranks),
                               "host_collectives_per_step": int(stats.host_collectives)}
            if ranks_view is not None:
                out["ranks"] = ranks_view
            if args.flux:
                out["scalar_flux_sum"] = float(sim.flux.sum().item())
            if world > 1:
                rec = one_rank_record(deck, nx, ntotal, K)
                if rec is None:
                    out["parity_vs_one_rank"] = {"recorded": False,
                                                 "note": "no one-rank record for this workload and "
                                                         "step count in profiles/one_rank_tally.json"}
                else:
                    out["parity_vs_one_rank"] = {
                        "recorded": True,
                        "event_counts_equal": all(tot[k] == rec["events"][k] for k in
                                                  ("facets", "collisions", "census", "histories")),
                        "global_tally_rel": abs(global_tally - rec["global_tally"]) /
                        abs(rec["global_tally"]),
                        "tolerance": 1e-12, "source": rec.get("source")}
            elif args.record_one_rank:
                path = one_rank_path()
                table = {"entries": []}
                if os.path.exists(path):
                    with open(path) as f:
                        table = json.load(f)
                table["entries"] = [e for e in table["entries"] if (e["deck"], e["nx"], e["nparticles"],
                                                                     e["steps"]) != (deck, nx, ntotal, K)]
                table["entries"].append({"deck": deck, "nx": nx, "nparticles": ntotal, "steps": K,
                                         "events": tot, "global_tally": global_tally,
                                         "source": f"bench.py --gpus 1 --steps {K} --workload "
                                                   f"{args.workload} --record-one-rank"})
                with open(path, "w") as f:
                    json.dump(table, f, indent=1)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"], oracle_run, n_sample = cpu_baseline(deck, nx, K,
                                                                         args.cpu_seconds, tmp)
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
                # "tally L2 vs omp3" (BASELINE.json metric): the HIP path on the CPU
                # baseline's own sample problem, per-cell tally against the oracle's
                sim.close()
                check = iface.Simulation(oracle_run.p, keys, values, device=local_rank,
                                         variant=args.variant)
                check.inject()
                ev = []
                for tt in range(1, K + 1):
                    r = check.step(tt)
                    ev.append((r.nprocessed, r.facets, r.collisions, r.census))
                t_gpu = check.tally_host()
                out["parity_vs_cpu"] = {
                    "sample_particles": n_sample,
                    "tally_l2_rel": float(np.linalg.norm(t_gpu - oracle_run.tally) /
                                          np.linalg.norm(oracle_run.tally)),
                    "tally_sum_rel": float(abs(t_gpu.sum() - oracle_run.tally.sum()) /
                                           abs(oracle_run.tally.sum())),
                    "event_counts_equal": ev == oracle_run.events,
                    "tolerance": 1e-6}
                check.close()
            print(json.dumps(out), flush=True)
        sim.close()  # (idempotent)
    if world > 1:
        lib.neutral_hip_comm_barrier()
        lib.neutral_hip_comm_stop()
        # RCCL was asked for and the ranks staged their exchange through the host instead: the
        # line above is a measurement of THAT, and must not pass for the RCCL figure
        if args.comm == "rccl" and transport != iface.COMM_RCCL:
            if rank == 0:
                print(f"bench.py: RCCL was asked for (--comm rccl) but the ranks fell back to the "
                      f"host route (transport {transport}): see the library's message above; "
                      f"the JSON line says so in ranks.transport_per_rank", file=sys.stderr,
                      flush=True)
            sys.exit(3)


if __name__ == "__main__":
    main()
