#!/bin/bash
# Throughput records of the two "next" rows that had none, results under gpurun_out/<tag>/:
#   flux_csp.json / flux_stream.json   bench lines with the scalar-flux tally kept (two 88-cell LDS windows), next to
#                                      plain lines of the same box, + kernel-trace stats of the csp one
#   decomposed_2x2.json                four ranks sharing this box's GPU, each owning a block of the mesh (host-staged
#                                      exchange: RCCL wants a GPU per rank): exchange rounds, emigrants per round, stage
#                                      times per rank; kernel-trace stats of rank 0 (pack / append kernels)
R=$GRAFT_REPO_ROOT; tag=${1:-rXX}; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in csp stream; do
  python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-lazy-leg > $out/plain_$w.json 2> $out/plain_$w.err || exit 1
  python3 $R/bench.py --workload $w --flux --steps 5 --warmup 1 --no-cpu-baseline --no-lazy-leg > $out/flux_$w.json 2> $out/flux_$w.err || exit 1
done
rocprofv3 --kernel-trace --stats -d $out/ktrace_flux --output-format csv -- python3 $R/bench.py --flux --steps 5 --warmup 0 --no-cpu-baseline --no-lazy-leg > $out/ktrace_flux.log 2>&1
for f in $out/ktrace_flux/*/*_kernel_stats.csv; do cp $f $out/kernel_stats_flux.csv; done
# decomposed 2x2 on the shared GPU: the ranks started by hand (rank 0 under the profiler: no launcher in between)
export WORLD_SIZE=4 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 NEUTRAL_COMM_PORT=29612 NEUTRAL_COMM_NONCE=$RANDOM$RANDOM HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--gpus 4 --decompose 2x2 --comm host --share-device --nparticles 20000000 --steps 5 --warmup 1 --no-cpu-baseline --no-lazy-leg"
for r in 1 2 3; do
  RANK=$r LOCAL_RANK=$r timeout -k 10 400 python3 $R/bench.py $ARGS > $out/decomposed_rank$r.log 2>&1 &
done
RANK=0 LOCAL_RANK=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/ktrace_decomposed --output-format csv -- python3 $R/bench.py $ARGS > $out/decomposed_2x2.json 2> $out/decomposed_2x2.err
wait
for f in $out/ktrace_decomposed/*/*_kernel_stats.csv; do cp $f $out/kernel_stats_decomposed_rank0.csv; done
find $out -name "*_kernel_trace.csv" -size +1M -delete
python3 - $out <<'PY'
import json, sys, os
out = sys.argv[1]
def line(f):
    try:
        return json.loads([l for l in open(os.path.join(out, f)).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return None
for w in ("csp", "stream"):
    p, f = line(f"plain_{w}.json"), line(f"flux_{w}.json")
    if p and f:
        ks = lambda d: {k["name"][:14]: round(k["ms_per_launch"], 2) for k in d["kernels"]}
        print(w, "plain", round(p["ms_per_step"], 2), ks(p), "| flux", round(f["ms_per_step"], 2), ks(f), "tile", f["tile_cells"], "flux sum", f.get("scalar_flux_sum"))
d = line("decomposed_2x2.json")
if d:
    print("decomposed 2x2:", round(d["ms_per_step"], 2), "ms/step", "%.3e" % d["value"], json.dumps(d["ranks"].get("decomposition")), "device ms", d["ranks"]["device_ms_per_step"])
PY
head -12 $out/kernel_stats_decomposed_rank0.csv | cut -c1-140
