#!/bin/bash
# ablation C: K2 pass policy thresholds
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids; }
for lib in "" cmin32 cmin56 cmin64 r4 r16; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 10000000 10 1
  run split 800 5000000 1 1
  run scatter 400 5000000 1 1
done
unset NEUTRAL_HIP_LIB
run split 800 5000000 1 0
run scatter 400 5000000 1 0
