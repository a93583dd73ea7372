#!/bin/bash
# tile edge of the tiled pipeline (NEUTRAL_TILE_CELLS: 16 default, 32, 64; the LDS window stays 128 cells):
# larger tiles hold more particles (sparse decks reach the window threshold) but leave less margin
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" tile32 tile64; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run stream 4000 1000000 1 2
  run csp 4000 1000000 10 2
  run split 4000 1000000 1 2
  run stream 400 10000000 1 2
  run csp 400 100000000 3 2
  run csp 400 12500000 10 2
done
