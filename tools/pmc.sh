#!/bin/bash
# PMC profile of the history kernels for one ablate.py workload.
#   tools/pmc.sh <tag> <deck> <nx> <n> <its> <variant>
# Two --pmc passes (8 SQ counters each), summaries in gpurun_out/pmc_<tag>.txt
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY"
B="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64"
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $A -d /tmp/pmc_$tag/a --output-format csv -- python3 $R/tools/ablate.py "$@" > /tmp/pmc_$tag.a.log 2>&1
rocprofv3 --pmc $B -d /tmp/pmc_$tag/b --output-format csv -- python3 $R/tools/ablate.py "$@" > /tmp/pmc_$tag.b.log 2>&1
python3 - "$tag" "$@" <<'PY' | tee $R/gpurun_out/pmc_$1.txt
import csv, glob, sys, collections
tag = sys.argv[1]
allk = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f'/tmp/pmc_{tag}/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for short in ('stream_kernel', 'history_regroup_kernel', 'history_kernel'):
            if short + '<' in r['Kernel_Name']:
                allk[short][r['Counter_Name']] += float(r['Counter_Value'])
print(tag, ' '.join(sys.argv[2:]))
for kname, tot in allk.items():
  print(' kernel', kname)
  for k in sorted(tot): print(f'  {k:28s} {tot[k]:.4e}')
  g = tot.get
  if g('SQ_ACTIVE_INST_VALU'):
      print('  VALU lane utilisation      %.1f %%' % (100*g('SQ_THREAD_CYCLES_VALU',0)/(g('SQ_ACTIVE_INST_VALU')*64)))
  if g('SQ_WAVE_CYCLES'):
      wc = g('SQ_WAVE_CYCLES')
      print('  of wave-cycles: wait_any %.1f %%  wait_inst_any %.1f %%' % (100*g('SQ_WAIT_ANY',0)/wc, 100*g('SQ_WAIT_INST_ANY',0)/wc))
      print('  VALU insts per wave-cycle(quad) %.3f' % (g('SQ_INSTS_VALU',0)/wc))
  if g('SQ_BUSY_CYCLES') and g('SQ_ACTIVE_INST_VALU'):
      print('  active_inst_valu / busy_cycles %.3f ; active_inst_any / busy %.3f' % (g('SQ_ACTIVE_INST_VALU')/g('SQ_BUSY_CYCLES'), g('SQ_ACTIVE_INST_ANY',0)/g('SQ_BUSY_CYCLES')))
PY
