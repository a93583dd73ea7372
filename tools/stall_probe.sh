#!/bin/bash
# Where a kernel's cycles go, beyond vector issue: PMC passes on one bench workload, summed per
# kernel (tools/stall_probe.py).  Results under gpurun_out/<tag>/stall/.
#   bash tools/stall_probe.sh <tag> [bench flags...]      e.g. --workload stream --steps 4
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; shift || true
out=$R/gpurun_out/$tag/stall; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P="--warmup 0 --no-cpu-baseline --no-lazy-leg $*"
pass() { name=$1; shift; rocprofv3 --pmc "$@" -d $out/pmc_$name --output-format csv -- python3 $R/bench.py $P > $out/pmc_$name.log 2>&1; echo "pmc $name done"; }
pass issue SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_ANY
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
python3 $R/tools/stall_probe.py $out/pmc_issue $out/pmc_lds $out/pmc_insts | tee $out/summary.txt
find $out -name "*_counter_collection.csv" -size +1M -delete
