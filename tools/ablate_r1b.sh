#!/bin/bash
# ablation B: K1 vs K2 (event-regrouped), refill thresholds, occupancy
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids; }
for v in 0 1; do
  echo "== variant $v"
  run stream 400 10000000 1 $v
  run csp 400 10000000 10 $v
  run scatter 400 2000000 1 $v
  run split 800 2000000 1 $v
done
for lib in refill8 refill16 refill32 refill48 k2w3 notally; do
  export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so
  run csp 400 10000000 10 1
  run split 800 2000000 1 1
  run stream 400 10000000 1 1
done
