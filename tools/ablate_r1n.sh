#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" rep16 rep32; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 4 2
  run stream 400 10000000 1 2
  run csp 400 12500000 10 2
done
