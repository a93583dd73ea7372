#!/bin/bash
# collision stage: equal shares against shares in proportion to what a wave is served (NEUTRAL_SHARE_WEIGHT),
# csp at the full size and at the 8-GPU share, same box (ms per launch of the collision stage)
mkdir -p gpurun_out/r04
for n in 100000000 12500000; do
  for w in "$@"; do
    NEUTRAL_SHARE_WEIGHT=$w timeout -k 10 200 python bench.py --nparticles $n --steps 10 --warmup 2 --no-cpu-baseline --no-lazy-leg > gpurun_out/r04/sw_${n}_w$w.json 2> gpurun_out/r04/sw_${n}_w$w.err || exit 1
    python - $n $w <<'PY'
import json, sys
n, w = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/r04/sw_{n}_w{w}.json").read().strip().splitlines()[-1])
print("particles", n, "weight", w, "ms/step", round(d["ms_per_step"], 3), {k["name"][:14]: round(k["ms_per_launch"], 3) for k in d["kernels"]}, "frac", round((d["roofline"] or {}).get("frac") or 0, 4))
PY
  done
done
