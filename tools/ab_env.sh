#!/bin/bash
# same-box A/B of environment settings on a bench workload, every setting twice, interleaved:
#   tools/ab_env.sh OUTDIR "BENCH ARGS" "NAME=VALUE ..." "NAME=VALUE ..." ...   ("-" = nothing set)
out=$1; shift; args=$1; shift
mkdir -p $out
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    [ "$setting" = "-" ] && setting=""
    env $setting timeout -k 10 300 python bench.py $args --no-cpu-baseline > $out/ab_${i}_$rep.json 2> $out/ab_${i}_$rep.err || { tail -5 $out/ab_${i}_$rep.err; exit 1; }
    echo "$setting" > $out/ab_${i}.setting
  done
done
python - $out <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/ab_*_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ks = {k['name'][:14]: round(k['ms_per_launch'], 2) for k in d['kernels']}
    lz = d.get('lazy_export') or {}
    alone = (d.get('roofline', {}).get('alone') or {}).get('kernel_ms_avg', 0)
    setting = open(f.rsplit('_', 1)[0] + '.setting').read().strip() or '(defaults)'
    print(f.split('/')[-1], setting, 'value %.4e' % d['value'], 'ms/step', round(d['ms_per_step'], 3), ks, 'lazy ms', round(lz.get('ms_per_step', 0), 3), 'stage alone', round(alone, 2))
PY
