#!/bin/bash
# End-of-round evidence: default bench (JSON, kernel-trace stats, PMC passes -> per-event
# coefficients), all BASELINE configurations per variant, the reference's decks as shipped,
# the bench workload's multi-GPU shares on one GPU, per-kernel traces and SQ counters of the
# other decks.
#   bash tools/profile_round.sh <tag>      (on the GPU box; results under gpurun_out/<tag>/)
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; out=$R/gpurun_out/$tag; mkdir -p $out
bash $R/tools/profile_bench.sh $tag
echo "profile_bench done" >> $out/progress.log
bash $R/tools/baseline_configs.sh > $out/baseline_configs.log 2>&1
echo "baseline_configs done" >> $out/progress.log
cd $R
python tools/ablate.py matrix --run "stream 4000 1000000 1 2" --run "csp 4000 1000000 10 2" \
  --run "scatter 4000 10000000 2 2" --run "split 4000 1000000 1 2" > $out/default_decks.log 2>&1
NEUTRAL_EAGER_EXPORT=1 python tools/ablate.py matrix --run "stream 4000 1000000 1 2" \
  --run "csp 4000 1000000 10 2" >> $out/default_decks.log 2>&1
echo "default decks done" >> $out/progress.log
bash tools/share_bench.sh $tag/share > $out/share_bench.log 2>&1
echo "shares done" >> $out/progress.log
for cfg in "stream 400 10000000 1" "scatter 400 20000000 1" "split 800 20000000 1"; do
  set -- $cfg
  bash $R/tools/ktrace.sh $1 $cfg 2 > $out/ktrace_$1.txt 2>&1
  bash $R/tools/pmc.sh $1 $cfg 2 > $out/pmc_sq_$1.txt 2>&1
  echo "$1 traces done" >> $out/progress.log
done
bash $R/tools/pmc.sh csp csp 400 20000000 10 2 > $out/pmc_sq_csp.txt 2>&1
tail -4 $out/baseline_configs.log
