#!/bin/bash
# empty lanes that trigger a refill in the stream kernel (NEUTRAL_STREAM_REFILL_MIN, default 32)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" sr16 sr20 sr24 sr28; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run stream 400 10000000 1 2
  run csp 400 100000000 3 2
done
