#!/bin/bash
# same-box A/B: the stream kernel at other workgroup shapes (builds: make -C neutral_amd variant TAG=... EXTRA=...)
out=gpurun_out/r04/occ; mkdir -p $out
for lib in "$@"; do
  if [ $lib = new ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lazy-leg > $out/csp_$lib.json 2> $out/csp_$lib.err || { tail -5 $out/csp_$lib.err; exit 1; }
  grep -q "Memory access fault" $out/csp_$lib.err && exit 9
  timeout -k 10 200 python bench.py --workload stream --steps 5 --warmup 1 --no-cpu-baseline --no-lazy-leg > $out/stream_$lib.json 2> $out/stream_$lib.err || { tail -5 $out/stream_$lib.err; exit 1; }
  grep -h "\[exp\]" $out/csp_$lib.err
done
python - "$@" <<'PY'
import json, sys
for lib in sys.argv[1:]:
    for w in ("csp", "stream"):
        d = json.loads(open(f"gpurun_out/r04/occ/{w}_{lib}.json").read().strip().splitlines()[-1])
        ks = {k['name'][:14]: round(k['ms_per_launch'], 2) for k in d['kernels']}
        print(lib, w, 'ms/step', round(d['ms_per_step'], 3), ks, 'passes', d.get('stream_passes_per_step'), 'tally', d.get('global_tally'))
PY
