#!/bin/bash
# instruction diet of the collision body: base = before, rotA = Threefry rotates as
# v_alignbit_b32 pairs, default = + number_density kept across collisions and the
# exact one-half shortcuts of identical tables
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in base rotA ""; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 10 2
  run scatter 400 20000000 1 2
  run stream 400 10000000 1 2
  run split 800 20000000 1 2
done
