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
# vector issue: wave-level VALU instructions (and the quarter-rate f64 ones), lane-cycles, and the
# busy-clock counter that gives the effective shader clock of each kernel (MI355X_MICROARCH.md, DVFS)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $out/pmc_valu --output-format csv -- python3 $R/bench.py --warmup 0 --no-cpu-baseline > $out/pmc_valu.log 2>&1
python3 $R/tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write csp 400 100000000 2 $out/pmc_valu | tee $out/pmc_traffic.log
cp $R/profiles/pmc_traffic.json $out/pmc_traffic.json
cat $out/ktrace/*/*_kernel_stats.csv | cut -c1-150 | head -8
