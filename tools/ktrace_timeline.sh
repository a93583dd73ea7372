#!/bin/bash
# One steady-state timestep of the bench workload as a timeline: every kernel of the step in launch order with its start
# (relative to the step's first kernel), duration and the gap before it.   tools/ktrace_timeline.sh [bench flags]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ktl
rocprofv3 --kernel-trace -d /tmp/ktl --output-format csv -- python3 $R/bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-lazy-leg "$@" > /tmp/ktl.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/ktl/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:]))
rows.sort()
# steps begin at a tables_check_kernel; take the fifth one
starts = [i for i, r in enumerate(rows) if 'tables_check_kernel' in r[2]]
a, b = starts[4], starts[5]
t0 = rows[a][0]; prev_end = t0
busy = 0
for s, e, n in rows[a:b]:
    print('%9.3f ms  %8.3f ms  gap %7.3f  %s' % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, n))
    prev_end = max(prev_end, e); busy += e - s
print('step: %.3f ms from the first kernel to the next step\'s first; sum of kernel durations %.3f ms (kernels overlap where streams do)' % ((rows[b][0] - t0) / 1e6, busy / 1e6))
PY
