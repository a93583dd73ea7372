#!/bin/bash
# End-of-round evidence: default bench (JSON, kernel-trace stats, FETCH/WRITE passes), all BASELINE
# configurations per variant, per-kernel traces and SQ counters of the other decks.
#   bash tools/profile_round.sh <tag>      (on the GPU box; results under gpurun_out/<tag>/)
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; out=$R/gpurun_out/$tag; mkdir -p $out
bash $R/tools/profile_bench.sh $tag
bash $R/tools/baseline_configs.sh > $out/baseline_configs.log 2>&1
for cfg in "stream 400 10000000 1" "scatter 400 20000000 1" "split 800 20000000 1"; do
  set -- $cfg
  bash $R/tools/ktrace.sh $1 $cfg 2 > $out/ktrace_$1.txt 2>&1
  bash $R/tools/pmc.sh $1 $cfg 2 > $out/pmc_sq_$1.txt 2>&1
done
bash $R/tools/pmc.sh csp csp 400 20000000 10 2 > $out/pmc_sq_csp.txt 2>&1
tail -4 $out/baseline_configs.log
