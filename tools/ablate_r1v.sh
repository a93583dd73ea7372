#!/bin/bash
# how many waiting histories switch the collision stage's time slicing on
# (NEUTRAL_SLICE_WINDOW): default 128, variants 64 / 256 / 512
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" sw64 sw256 sw512; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 10 2
  run csp 400 12500000 10 2
  run scatter 400 20000000 1 2
  run split 800 20000000 1 2
done
