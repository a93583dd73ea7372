#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 120 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" ld32 ld16 ld8; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 3 2
  run stream 400 10000000 1 2
done
