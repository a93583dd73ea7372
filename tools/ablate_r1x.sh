#!/bin/bash
# largest share per wave that is still pooled and time-sliced (NEUTRAL_POOL_MAX_SHARE)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" pm4096 pm8192; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 10 2
  run scatter 400 12500000 1 2
  run scatter 400 20000000 1 2
  run split 800 20000000 1 2
  run split 800 12500000 1 2
done
