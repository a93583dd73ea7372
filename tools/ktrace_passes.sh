#!/bin/bash
# Per-launch durations of stream_kernel in launch order (the passes of a step): tools/ktrace_passes.sh <workload> <steps>
R=$GRAFT_REPO_ROOT; w=${1:-stream4000}; steps=${2:-1}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ktp
rocprofv3 --kernel-trace -d /tmp/ktp --output-format csv -- python3 $R/bench.py --workload $w --steps $steps --warmup 0 --no-cpu-baseline --no-lazy-leg > /tmp/ktp.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/ktp/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'stream_kernel' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
rows.sort()
d = [x[1] for x in rows]
print(len(d), 'launches; ms each:', ' '.join('%.2f' % x for x in d))
print('total %.2f ms; first 8: %.2f; first 16: %.2f; first 32: %.2f' % (sum(d), sum(d[:8]), sum(d[:16]), sum(d[:32])))
PY
