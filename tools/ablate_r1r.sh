#!/bin/bash
# time-slice length of the collision stage (NEUTRAL_SLICE_PASSES), csp at the
# 1-GPU and 8-GPU shares
cd $GRAFT_REPO_ROOT
run() { timeout 120 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" sl16 sl32 sl128 sl256; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 10 2
  run csp 400 12500000 10 2
  run split 800 12500000 1 2
done
