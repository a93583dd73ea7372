#!/bin/bash
# The bench lines of the headline workload again, priced with the coefficients committed by tools/profile_round.sh <tag> headline
# (bench.json with the CPU baseline, bench_priced.json, bench_driver_style.json = --steps 20 --warmup 5), then every other workload's.
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
python3 bench.py --no-cpu-baseline > $out/bench_priced.json 2> $out/bench_priced.err || exit 1
python3 bench.py --steps 20 --warmup 5 --record-one-rank > $out/bench_driver_style.json 2> $out/bench_driver_style.err || exit 1
python3 bench.py --steps 10 --warmup 1 --record-one-rank --no-cpu-baseline --no-lazy-leg > /dev/null 2>&1
cp profiles/one_rank_tally.json $out/one_rank_tally.json
bash tools/profile_round.sh $tag priced
