#!/bin/bash
# End-of-round evidence, in parts that each fit one gpurun call (results under gpurun_out/<tag>/):
#   bash tools/profile_round.sh <tag> headline   default bench (JSON, kernel-trace stats, PMC passes -> per-event
#                                                coefficients, priced again), the driver-style run (--steps 20
#                                                --warmup 5) that also records the one-rank tallies N > 1 runs check
#   bash tools/profile_round.sh <tag> configs    light profile (bench line, kernel-trace stats, PMC coefficients)
#                                                of the other BASELINE configurations
#   bash tools/profile_round.sh <tag> shipped    the same for the reference's decks as shipped (4000^2)
#   bash tools/profile_round.sh <tag> matrix     all BASELINE configurations per kernel variant, the shipped decks,
#                                                the bench workload's multi-GPU shares on one GPU
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; part=${2:-headline}; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
case $part in
headline)
  bash tools/profile_bench.sh $tag
  python3 bench.py --steps 20 --warmup 5 --record-one-rank > $out/bench_driver_style.json 2> $out/bench_driver_style.err
  python3 bench.py --steps 10 --warmup 1 --record-one-rank --no-cpu-baseline --no-lazy-leg > /dev/null 2>&1
  tail -c 300 $out/bench_driver_style.json; echo
  cp profiles/one_rank_tally.json $out/one_rank_tally.json
  ;;
configs)
  bash tools/profile_bench.sh $tag/stream --light --workload stream --steps 10 --warmup 1
  bash tools/profile_bench.sh $tag/scatter --light --workload scatter --steps 2 --warmup 0
  bash tools/profile_bench.sh $tag/split --light --workload split --steps 1 --warmup 0
  ;;
shipped)
  bash tools/profile_bench.sh $tag/stream4000 --light --workload stream4000 --steps 4 --warmup 1
  bash tools/profile_bench.sh $tag/csp4000 --light --workload csp4000 --steps 10 --warmup 1
  ;;
share)
  # the coefficients of what an 8-GPU rank holds of the bench workload (priced into the N = 8 line)
  SHARE=8 bash tools/profile_bench.sh $tag/share8 --light --nparticles 12500000 --steps 20 --warmup 5
  ;;
priced)
  # every workload's bench line again, priced with the coefficients committed by the parts above
  for w in "stream 10 1" "scatter 2 0" "split 1 0" "stream4000 4 1" "csp4000 10 1"; do
    set -- $w
    mkdir -p $out/$1; python3 bench.py --workload $1 --steps $2 --warmup $3 --no-cpu-baseline > $out/$1/bench_priced.json 2> $out/$1/bench_priced.err
    tail -c 200 $out/$1/bench_priced.json; echo
  done
  ;;
matrix)
  bash tools/baseline_configs.sh > $out/baseline_configs.log 2>&1
  python tools/ablate.py matrix --run "stream 4000 1000000 1 2" --run "csp 4000 1000000 10 2" \
    --run "scatter 4000 10000000 2 2" --run "split 4000 1000000 1 2" > $out/default_decks.log 2>&1
  NEUTRAL_EAGER_EXPORT=1 python tools/ablate.py matrix --run "stream 4000 1000000 1 2" \
    --run "csp 4000 1000000 10 2" >> $out/default_decks.log 2>&1
  bash tools/share_bench.sh $tag/share > $out/share_bench.log 2>&1
  bash tools/ktrace.sh share8 csp 400 12500000 10 2 > $out/ktrace_share8.txt 2>&1
  tail -4 $out/baseline_configs.log; cat $out/default_decks.log | cut -c1-160; tail -5 $out/share_bench.log
  ;;
esac
