#!/usr/bin/env python3
"""Folds rocprofv3 --pmc passes of `bench.py` into profiles/pmc_per_event.json, which
bench.py reads back for its vector-issue roofline and its HBM traffic figures.

  python tools/pmc_events.py --bench <bench.json of a run with the same flags> \
      --source "<what was profiled>" <pass_dir> [<pass_dir> ...]

Every pass is the SAME command (bench.py --warmup 0 --no-cpu-baseline --no-lazy-leg)
under `rocprofv3 --pmc <counters>`: the launches it profiles are exactly the timed
launches whose event counts the bench line reports (`kernels[].events_per_launch`).
Counters are summed over all dispatches of a kernel and divided by the kernel's
primary event count (facets for the stream kernel, collisions for the collision
kernel): counters PER EVENT, which bench.py multiplies by the events of whatever run
it is pricing -- the instruction counts then match the run being timed, not the run
that was profiled.

HBM bytes per event = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / events: FETCH_SIZE and
WRITE_SIZE are in KB, collected in separate passes, and on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section).  That
correction is calibrated for 16-B-per-lane streams; this path reads 8 B per lane and
gathers 80-B records, so the read side is an upper estimate.
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("history_kernel", "history_regroup_kernel", "stream_kernel")
PRIMARY = {"stream_kernel": "facets", "history_regroup_kernel": "collisions",
           "history_kernel": "collisions"}


def fold(dirs):
    """{kernel: {counter: sum over dispatches}}, {kernel: dispatches}"""
    sums, launches = {}, {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            seen = {}
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                for short in KERNELS:
                    if short + "<" in name or short + "(" in name:
                        c = r["Counter_Name"]
                        sums.setdefault(short, {}).setdefault(c, 0.0)
                        sums[short][c] += float(r["Counter_Value"])
                        seen.setdefault((short, c), set()).add(r["Dispatch_Id"])
            for (short, c), ids in seen.items():
                launches[short] = max(launches.get(short, 0), len(ids))
    return sums, launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", required=True)
    ap.add_argument("--source", required=True)
    ap.add_argument("--share", type=int, default=None,
                    help="the profiled run is the share one of this many ranks holds")
    ap.add_argument("dirs", nargs="+")
    a = ap.parse_args()
    line = [ln for ln in open(a.bench).read().splitlines() if ln.startswith("{")][-1]
    bench = json.loads(line)
    cfg = bench["config"]
    sums, launches = fold(a.dirs)
    path = os.path.join(ROOT, "profiles", "pmc_per_event.json")
    table = {"entries": []}
    if os.path.exists(path):
        table = json.load(open(path))
    for k in bench["kernels"]:
        name = k["name"]
        if name not in sums:
            continue
        events = k["events_per_launch"][PRIMARY[name]] * bench["steps"]
        if events <= 0:
            continue
        per = {c: v / events for c, v in sums[name].items()}
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            per["hbm_bytes"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
        e = {"deck": cfg["deck"], "nx": cfg["nx"], "variant": cfg["kernel_variant"],
             "kernel": name, "event": PRIMARY[name], "events_profiled": events,
             "nparticles_profiled": cfg["nparticles"], "steps_profiled": bench["steps"],
             "dispatches_profiled": launches.get(name), "source": a.source, "per_event": per,
             "kernel_ms_per_launch_same_flags": k.get("ms_per_launch")}
        passes = ((k.get("valu_issue") or {}).get("collision_passes_per_launch") or 0) * bench["steps"]
        if passes and "SQ_INSTS_VALU" in sums[name]:
            # what a wave-level collision pass issues on average, refills and hand-backs included:
            # the DYNAMIC counterpart of the static trip of tools/isa_histogram.py (which also counts
            # the instructions of blocks a trip rarely enters)
            e["valu_insts_per_collision_pass"] = sums[name]["SQ_INSTS_VALU"] / passes
            e["collision_passes_profiled"] = passes
        if a.share:
            e["share_of"] = a.share   # (coefficients of a multi-GPU rank's share of the workload)
        table["entries"] = [x for x in table["entries"]
                            if (x["deck"], x["nx"], x["variant"], x["kernel"], x.get("share_of")) !=
                            (e["deck"], e["nx"], e["variant"], e["kernel"], e.get("share_of"))]
        table["entries"].append(e)
        print(name, "events", events, {c: round(v, 4) for c, v in sorted(per.items())})
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
