#!/bin/bash
# same-box A/B of round 4: tile queues on/off on the stream-bound workloads, this build against the round-3 library on csp
mkdir -p gpurun_out/r04
for w in stream stream4000 csp4000; do
  for q in 1 0; do
    NEUTRAL_STREAM_QUEUES=$q timeout -k 10 200 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r04/ab_${w}_q$q.json 2> gpurun_out/r04/ab_${w}_q$q.err || exit 1
    grep -q "Memory access fault" gpurun_out/r04/ab_${w}_q$q.err && exit 9
  done
done
i=0
for lib in r03 new r03 new; do
  i=$((i+1))
  if [ $lib = r03 ]; then export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_r03.so; else unset NEUTRAL_HIP_LIB; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04/ab_csp_${lib}_$i.json 2> gpurun_out/r04/ab_csp_${lib}_$i.err || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    ks = {k['name'][:14]: round(k['ms_per_launch'], 2) for k in d['kernels']}
    lz = d.get('lazy_export') or {}
    print(f.split('/')[-1], 'ms/step', round(d['ms_per_step'], 3), 'value %.4g' % d['value'], ks, 'passes', d.get('stream_passes_per_step'), 'lazy', round(lz.get('ms_per_step', 0), 2), 'wb', round(lz.get('ms_per_writeback_on_demand', 0), 2), 'frac', (d['roofline'] or {}).get('frac'))
PY
