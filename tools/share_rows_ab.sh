#!/bin/bash
# the 8- and 4-GPU shares of the bench workload with the collision stage held to fewer workgroups (NEUTRAL_K2_MAX_BLOCKS)
out=gpurun_out/r04/rows; mkdir -p $out
for n in 12500000 25000000; do
  for mb in 0 512 768; do
    NEUTRAL_K2_MAX_BLOCKS=$mb timeout -k 10 200 python bench.py --nparticles $n --no-cpu-baseline > $out/s_${n}_$mb.json 2> $out/s_${n}_$mb.err || exit 1
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04/rows/s_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['ms_per_step'], 3), 'lazy', round(d['lazy_export']['ms_per_step'], 3), {k['name'][:12]: round(k['ms_per_launch'], 2) for k in d['kernels']}, 'alone', round((d['roofline'].get('alone') or {}).get('kernel_ms_avg', 0), 2))
PY
