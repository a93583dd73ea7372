#!/usr/bin/env python3
"""Table of the history kernels' register use from `make -C neutral_amd asm`
(neutral_amd/build/resource_usage.txt): VGPRs, SGPRs, scratch, waves per SIMD.
  python tools/kernel_resources.py [pattern]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TXT = os.path.join(ROOT, "neutral_amd", "build", "resource_usage.txt")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", o).replace("neutral::", "").replace("void ", "") for o in out]


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "history|stream_kernel"
    rows, cur = [], None
    for line in open(TXT):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, rx in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                        ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(rx, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    rows = [r for r in rows if re.search(pat, r["name"])]
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        print(f"{n:70s} vgpr {r.get('vgpr', 0):3d} agpr {r.get('agpr', 0):3d} sgpr {r.get('sgpr', 0):3d} "
              f"scratch {r.get('scratch', 0):4d} waves/SIMD {r.get('occ', 0)}")


if __name__ == "__main__":
    main()
