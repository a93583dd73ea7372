#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel (short name) over the directories given and prints
them next to each other, with each counter also as a ratio to SQ_INSTS_VALU (tools/stall_probe.sh)."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def main():
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for d in sys.argv[1:]:
        seen = set()
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    key = (k, r.get("Dispatch_Id"))
                    if key not in seen:
                        seen.add(key)
        for k, _ in seen:
            launches[k] += 1
    for k in sorted(sums, key=lambda k: -sums[k].get("SQ_INSTS_VALU", 0.0)):
        c = sums[k]
        valu = c.get("SQ_INSTS_VALU", 0.0)
        if valu < 1e6:
            continue
        print(f"== {k}")
        for name in sorted(c):
            ratio = f"  {c[name] / valu:10.4f} per VALU instruction" if valu else ""
            print(f"  {name:28s} {c[name]:16.4e}{ratio}")


if __name__ == "__main__":
    main()
