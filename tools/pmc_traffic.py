#!/usr/bin/env python3
"""Folds rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into
profiles/pmc_traffic.json (read back by bench.py for roofline.traffic).

  python tools/pmc_traffic.py <fetch_dir> <write_dir> <deck> <nx> <nparticles> <variant> [<valu_dir>]

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE
are in KB and on gfx950 FETCH_SIZE reads half of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section).  The x2 is calibrated for 16 B/lane loads;
this path reads 8 B/lane, so the read side is an upper estimate.

With <valu_dir> (a pass of SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU
SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE) each entry also gets the
wave-level vector instructions per launch and the busy-clock count, from which
bench.py prints the kernel's vector-issue rate next to the HBM roofline.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"history_kernel": "history_kernel", "history_regroup_kernel": "history_regroup_kernel",
           "stream_kernel": "stream_kernel"}


def per_kernel(directory, counter):
    out = {}
    for f in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for short in KERNELS:
                if short + "<" in r["Kernel_Name"] or short + "(" in r["Kernel_Name"]:
                    out.setdefault(short, []).append(float(r["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, deck, nx, n, variant = sys.argv[1:7]
    valu_dir = sys.argv[7] if len(sys.argv) > 7 else None
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    valu = {}
    if valu_dir:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F64", "SQ_ACTIVE_INST_VALU",
                  "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"):
            valu[c] = per_kernel(valu_dir, c)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = {"entries": []}
    if os.path.exists(path):
        table = json.load(open(path))
    for k in sorted(set(fetch) & set(write)):
        f = sum(fetch[k]) / len(fetch[k])
        w = sum(write[k]) / len(write[k])
        e = {"deck": deck, "nx": int(nx), "nparticles": int(n), "variant": int(variant), "kernel": k,
             "launches": len(fetch[k]), "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
             "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
        for c, per in valu.items():
            if k in per:
                e[c + "_per_launch"] = sum(per[k]) / len(per[k])
        table["entries"] = [x for x in table["entries"]
                            if (x["deck"], x["nx"], x["nparticles"], x["variant"], x["kernel"]) !=
                            (e["deck"], e["nx"], e["nparticles"], e["variant"], e["kernel"])]
        table["entries"].append(e)
        print(e)
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
