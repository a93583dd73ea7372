#!/bin/bash
# same-box A/B of round 4 (results under gpurun_out/r04/): the stream-bound workloads with the tile
# queues on and off, and csp with this build, an A/B build and the round-3 library in turn
mkdir -p gpurun_out/r04
for w in stream4000 csp4000 stream; do
  for q in 1 0; do
    NEUTRAL_STREAM_QUEUES=$q timeout -k 10 200 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-lazy-leg > gpurun_out/r04/ab_${w}_q$q.json 2> gpurun_out/r04/ab_${w}_q$q.err || exit 1
    grep -q "Memory access fault" gpurun_out/r04/ab_${w}_q$q.err && exit 9
  done
done
i=0
for lib in "$@"; do
  i=$((i+1))
  if [ $lib = new ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lazy-leg > gpurun_out/r04/ab_csp_${lib}_$i.json 2> gpurun_out/r04/ab_csp_${lib}_$i.err || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    ks = {k['name'][:14]: round(k['ms_per_launch'], 2) for k in d['kernels']}
    q = d.get('stream_queue') or {}
    print(f.split('/')[-1], 'ms/step', round(d['ms_per_step'], 3), ks, 'passes', d.get('stream_passes_per_step'),
          {k[:-9]: int(v) for k, v in q.items()})
PY
