#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" crep1 crep4 rep8 rep16; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 10 2
  run scatter 400 5000000 1 2
  run split 800 5000000 1 2
done
