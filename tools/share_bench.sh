#!/bin/bash
# csp at the 8-, 4-, 2- and 1-GPU shares of the bench workload on ONE GPU: the
# single-GPU evidence for strong scaling.
# usage: tools/share_bench.sh <tag> [sizes...]
set -e
tag=${1:-share}; shift || true
sizes=${@:-12500000 25000000 50000000 100000000}
mkdir -p gpurun_out/$(dirname $tag)
for n in $sizes; do
  python bench.py --nparticles $n --no-cpu-baseline > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err
done
python - "$tag" $sizes <<'PY'
import json, sys
tag = sys.argv[1]
for n in sys.argv[2:]:
    d = json.loads(open(f"gpurun_out/{tag}_{n}.json").read().strip().splitlines()[-1])
    print(n, "default ABI", round(d["ms_per_step"], 3), "ms/step", "%.3e" % d["value"],
          "| lazy export", round(d["lazy_export"]["ms_per_step"], 3), "ms/step",
          "%.3e" % d["lazy_export"]["value"],
          [(k["name"][:14], round(k["ms_per_launch"], 2)) for k in d["kernels"]])
PY
