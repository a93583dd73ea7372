#!/bin/bash
# Profile of one bench.py workload on the GPU box: the bench line itself, rocprofv3 kernel-trace
# stats of the same command, and the separate PMC passes that tools/pmc_events.py folds into
# profiles/pmc_per_event.json (what bench.py prices its roofline with).
#   bash tools/profile_bench.sh <round-tag> [--light] [bench flags...]
#     --light: no L2 pass, no second priced run (the non-headline workloads)
# Results under gpurun_out/<round-tag>/.
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; shift || true
light=0; if [ "$1" = "--light" ]; then light=1; shift; fi
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json; echo
P="--warmup 0 --no-cpu-baseline --no-lazy-leg $*"
python3 $R/bench.py $P > $out/bench_profiled_flags.json 2> $out/bench_profiled_flags.err
rocprofv3 --kernel-trace --stats -d $out/ktrace --output-format csv -- python3 $R/bench.py $P > $out/ktrace.log 2>&1
pass() { name=$1; shift; rocprofv3 --pmc "$@" -d $out/pmc_$name --output-format csv -- python3 $R/bench.py $P > $out/pmc_$name.log 2>&1; echo "pmc $name done" >> $out/progress.log; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# wave-level vector instructions by class
pass valu SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
# occupancy of the issue slots, lane utilisation, stalls, busy clock
pass sq SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
dirs="$out/pmc_fetch $out/pmc_write $out/pmc_valu $out/pmc_sq"
if [ $light = 0 ]; then
  # L2: hits, misses, requests, atomics (L2 hit rate, atomic counts)
  pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
  dirs="$dirs $out/pmc_l2"
fi
# (SHARE=N in the environment: the run is the share one of N ranks holds -- its coefficients are kept
#  next to the whole workload's and priced into the N-GPU line)
python3 $R/tools/pmc_events.py --bench $out/bench_profiled_flags.json ${SHARE:+--share $SHARE} \
  --source "tools/profile_bench.sh $tag: bench.py $P under rocprofv3 --pmc" $dirs | tee $out/pmc_per_event.log
cp $R/profiles/pmc_per_event.json $out/pmc_per_event.json
for f in $out/ktrace/*/*_kernel_stats.csv; do cp $f $out/kernel_stats.csv; done
cut -c1-150 $out/kernel_stats.csv | head -10
if [ $light = 0 ]; then
  # the bench line again, now priced with this round's coefficients
  python3 $R/bench.py --no-cpu-baseline "$@" > $out/bench_priced.json 2> $out/bench_priced.err; tail -c 300 $out/bench_priced.json; echo
fi
# keep the raw per-dispatch CSVs out of the merge-back (tens of MB); summaries stay
find $out -name "*_counter_collection.csv" -size +1M -delete
find $out -name "*_kernel_trace.csv" -size +1M -delete
