#!/bin/bash
# Round profile of the default bench.py run (csp 400^2, 1e8 particles, tiled):
#   bench JSON, rocprofv3 kernel-trace stats, separate FETCH_SIZE / WRITE_SIZE passes.
# Usage on the GPU box: bash tools/profile_bench.sh <round-tag>
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json; echo
rocprofv3 --kernel-trace --stats -d $out/ktrace --output-format csv -- python3 $R/bench.py --warmup 0 --no-cpu-baseline > $out/ktrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $R/bench.py --warmup 0 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $R/bench.py --warmup 0 --no-cpu-baseline > $out/pmc_write.log 2>&1
cat $out/ktrace/*/*_kernel_stats.csv | cut -c1-150 | head -8
