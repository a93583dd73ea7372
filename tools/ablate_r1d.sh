#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids; }
for v in 1 0; do
  run csp 400 10000000 10 $v
  run split 800 5000000 1 $v
  run scatter 400 5000000 1 $v
  run stream 400 10000000 1 $v
done
export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_k1w2.so
run split 800 5000000 1 0
run scatter 400 5000000 1 0
