#!/bin/bash
# Why do the waves of a CU lose ~14 % when their histories mix (CU pools, levelling)?  The collision
# stage at the 8-GPU share with CU pools and with per-wave rings under the same PMC passes.
R=$GRAFT_REPO_ROOT; tag=${1:-r05}; out=$R/gpurun_out/$tag/pool_probe; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P="--warmup 0 --steps 8 --nparticles 25000000 --no-cpu-baseline --no-lazy-leg"
for mode in 1 0; do
  export NEUTRAL_CU_POOLS=$mode
  pass() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" -d $out/m${mode}_$name --output-format csv -- python3 $R/bench.py $P > $out/m${mode}_$name.log 2>&1; echo "mode $mode pass $name rc $?"; }
  pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
  pass lat SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  echo "== NEUTRAL_CU_POOLS=$mode" ; python3 $R/tools/stall_probe.py $out/m${mode}_sq $out/m${mode}_l2 $out/m${mode}_lat $out/m${mode}_fetch $out/m${mode}_write | sed -n '/history_regroup/,/^==/p' | head -40
done 2>&1 | tee $out/summary.txt
find $out -name "*_counter_collection.csv" -size +1M -delete
