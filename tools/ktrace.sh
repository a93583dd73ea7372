#!/bin/bash
# per-kernel time breakdown of one ablate.py workload:  tools/ktrace.sh <tag> <ablate args...>
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag --output-format csv -- python3 $R/tools/ablate.py "$@" > /tmp/kt_$tag.log 2>&1
grep -v amdgpu.ids /tmp/kt_$tag.log | tail -1
python3 - $tag <<'PY' | tee $R/gpurun_out/ktrace_$1.txt
import csv, glob, sys
f = glob.glob(f'/tmp/kt_{sys.argv[1]}/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('void neutral::','').replace('neutral::','')[:44]
    print(f"  {n:44s} calls {int(r['Calls']):4d}  total {float(r['TotalDurationNs'])/1e6:10.2f} ms  avg {float(r['AverageNs'])/1e6:9.3f} ms  {float(r['Percentage']):6.2f} %")
PY
