#!/bin/bash
# Round 5, verdict item 2 (co-residency): what does a collision stage of ONE wave per SIMD get done
# beside a stream kernel of three, and what does it cost the stream kernel?  Real kernels, two
# processes on the one GPU: A = the stream deck (stream kernel only) with 768-thread workgroups,
# B = the scatter deck (collision stage only) capped at one 256-thread workgroup per CU with the
# 17-KB index.  Each alone, then both at once.  Results: gpurun_out/<tag>/coresident/
tag=${1:-r05}; out=gpurun_out/$tag/coresident; mkdir -p $out
A="env NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_s768.so python bench.py --workload stream --steps 300 --warmup 2 --no-cpu-baseline --no-lazy-leg"
A4="python bench.py --workload stream --steps 300 --warmup 2 --no-cpu-baseline --no-lazy-leg"
B="env NEUTRAL_K2_MAX_BLOCKS=256 NEUTRAL_NO_FINE_INDEX=1 python bench.py --workload scatter --steps 1 --warmup 0 --no-cpu-baseline --no-lazy-leg"
B4="env NEUTRAL_NO_FINE_INDEX=1 python bench.py --workload scatter --steps 1 --warmup 0 --no-cpu-baseline --no-lazy-leg"
timeout -k 10 200 $A4 > $out/A_1024_alone.json 2> $out/A_1024_alone.err || exit 1
timeout -k 10 200 $A > $out/A_alone.json 2> $out/A_alone.err || exit 1
timeout -k 10 200 $B4 > $out/B_full_alone.json 2> $out/B_full_alone.err || exit 1
timeout -k 10 200 $B > $out/B_alone.json 2> $out/B_alone.err || exit 1
# both at once: A's 300 timed steps last ~5.5 s; B (start-up ~4 s, then one collision stage of ~1.5 s
# at one workgroup per CU) is started so that its stage falls inside them
( timeout -k 10 300 $A > $out/A_beside.json 2> $out/A_beside.err ) &
pa=$!
sleep 2
timeout -k 10 300 $B > $out/B_beside.json 2> $out/B_beside.err
wait $pa
python - $out <<'PY'
import json, sys
out = sys.argv[1]
def line(name):
    d = json.loads(open(f"{out}/{name}.json").read().strip().splitlines()[-1])
    ks = {k['name'][:14]: round(k['ms_per_launch'], 3) for k in d['kernels']}
    return d['ms_per_step'], ks, d['events']
res = {}
for name in ("A_1024_alone", "A_alone", "A_beside", "B_full_alone", "B_alone", "B_beside"):
    ms, ks, ev = line(name)
    res[name] = (ms, ks)
    print(f"{name:14s} ms/step {ms:10.3f}  {ks}  collisions {ev['collisions']:.3e} facets {ev['facets']:.3e}")
steps = 300
b_alone, b_beside = res["B_alone"][1]["history_regrou"], res["B_beside"][1]["history_regrou"]
a_extra = (res["A_beside"][0] - res["A_alone"][0]) * steps
print(f"B's stage: {b_alone:.1f} ms alone at one workgroup per CU, {b_beside:.1f} ms beside A ({b_beside / b_alone:.2f} x)")
print(f"A: {a_extra:.1f} ms longer over its {steps} steps with B's stage somewhere inside them "
      f"(= {a_extra / b_beside:.2f} of that stage's duration)")
PY
