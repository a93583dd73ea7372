#!/bin/bash
# same-box A/B of library builds on a bench workload: tools/ab_libs.sh OUTDIR "BENCH ARGS" lib1 lib2 ...
# (lib = "new" for the in-tree library, else the TAG of neutral_amd/build/libneutral_hip_TAG.so); every
# library runs twice, interleaved
out=$1; shift; args=$1; shift
mkdir -p $out
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    if [ $lib = new ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
    timeout -k 10 300 python bench.py $args --no-cpu-baseline > $out/ab_${lib}_$rep.json 2> $out/ab_${lib}_$rep.err || { tail -5 $out/ab_${lib}_$rep.err; exit 1; }
  done
done
python - $out <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/ab_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ks = {k['name'][:14]: round(k['ms_per_launch'], 2) for k in d['kernels']}
    lz = d.get('lazy_export') or {}
    print(f.split('/')[-1], 'value %.4e' % d['value'], 'ms/step', round(d['ms_per_step'], 3), ks, 'lazy ms', round(lz.get('ms_per_step', 0), 3))
PY
