#!/bin/bash
# Where the waves of the history kernels wait: one --pmc pass of the LDS / memory / scalar
# issue counters for one ablate.py workload.
#   tools/pmc_wait.sh <tag> <deck> <nx> <n> <its> <variant>     -> gpurun_out/pmc_wait_<tag>.txt
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT"
D="SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_IFETCH"
rm -rf /tmp/pmcw_$tag
rocprofv3 --pmc $C -d /tmp/pmcw_$tag/c --output-format csv -- python3 $R/tools/ablate.py "$@" > /tmp/pmcw_$tag.c.log 2>&1
rocprofv3 --pmc $D -d /tmp/pmcw_$tag/d --output-format csv -- python3 $R/tools/ablate.py "$@" > /tmp/pmcw_$tag.d.log 2>&1
python3 - "$tag" "$@" <<'PY' | tee $R/gpurun_out/pmc_wait_$tag.txt
import csv, glob, sys, collections
tag = sys.argv[1]
allk = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f'/tmp/pmcw_{tag}/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for short in ('stream_kernel', 'history_regroup_kernel', 'history_kernel'):
            if short + '<' in r['Kernel_Name']:
                allk[short][r['Counter_Name']] += float(r['Counter_Value'])
print(tag, ' '.join(sys.argv[2:]))
for kname, tot in allk.items():
    print(' kernel', kname)
    for k in sorted(tot):
        print(f'  {k:28s} {tot[k]:.4e}')
    wc = tot.get('SQ_WAVE_CYCLES')
    if wc:
        for k in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM',
                  'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC'):
            if k in tot:
                print(f'  {k} / wave-cycles = {100 * tot[k] / wc:.1f} %')
PY
