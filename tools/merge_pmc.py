#!/usr/bin/env python3
"""The parts of tools/profile_round.sh run in separate gpurun calls, each on a fresh copy of the tree: every
part's gpurun_out/<tag>/**/pmc_per_event.json is the committed table plus ITS new entries.  This folds the
new entries of all parts (source names the tag) into profiles/pmc_per_event.json.
    python tools/merge_pmc.py r05"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def key(e):
    return (e["deck"], e["nx"], e["variant"], e["kernel"], bool(e.get("flux", False)), e.get("share_of"))


path = os.path.join(ROOT, "profiles", "pmc_per_event.json")
table = {key(e): e for e in json.load(open(path))["entries"]}
# (every part's file is the committed table plus ITS new entries -- and the committed table may hold entries of the
#  same tag from an earlier run of the round: a file only contributes the entries made in its own directory,
#  "<tag>:" at the top, "<tag>/<dir>:" below)
top = os.path.join(ROOT, "gpurun_out", tag)
for f in sorted(glob.glob(os.path.join(top, "**", "pmc_per_event.json"), recursive=True)):
    sub = os.path.relpath(os.path.dirname(f), top)
    own = f" {tag}:" if sub == "." else f" {tag}/{sub}:"
    for e in json.load(open(f))["entries"]:
        if own in e.get("source", ""):
            table[key(e)] = e
out = {"entries": sorted(table.values(), key=lambda e: (e["deck"], e["nx"], e["kernel"], e.get("share_of") or 0))}
json.dump(out, open(path, "w"), indent=1)
for e in out["entries"]:
    print(e["deck"], e["nx"], e["kernel"], "share_of", e.get("share_of"), "VALU/event",
          round(e["per_event"].get("SQ_INSTS_VALU", 0), 4), "|", e["source"][:60])
