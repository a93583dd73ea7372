#!/bin/bash
# Drop-in proof: compiles the reference's UNMODIFIED driver and loader
# (main.c, neutral_data.c) with -DSoA against this repo's arch-compatible host
# headers (neutral_amd/host/*.h) and links them against libneutral_hip.so ALONE.
# Nothing of the reference is copied: sources are compiled where they lie, the
# binary lands in integration/_dropin/ (git-ignored; travels to the GPU box).
#
# This is an integration artefact, NOT the parity oracle.
set -euo pipefail
REFERENCE=${REFERENCE:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/integration/_dropin
if [ ! -f "$REFERENCE/main.c" ]; then
  echo "reference tree absent: keeping prebuilt $OUT (if any)"; exit 0
fi
if [ ! -f "$ROOT/neutral_amd/libneutral_hip.so" ]; then
  echo "libneutral_hip.so missing: run make -C neutral_amd first" >&2; exit 1
fi
mkdir -p "$OUT"
# The reference includes "../comms.h" etc. (main.c:1-5): the headers of the
# directory ABOVE it.  -I<dir>/a makes "../x.h" resolve to <dir>/x.h.
INC=$(mktemp -d)
trap 'rm -rf "$INC"' EXIT
mkdir -p "$INC/a"
for h in "$ROOT"/neutral_amd/host/*.h; do ln -s "$h" "$INC/$(basename "$h")"; done
CFLAGS="-std=gnu99 -O2 -fopenmp -DSoA -D__STDC_CONSTANT_MACROS -w -I$INC/a"
gcc $CFLAGS -c "$REFERENCE/main.c" -o "$OUT/main.o"
gcc $CFLAGS -c "$REFERENCE/neutral_data.c" -o "$OUT/neutral_data.o"
# $ORIGIN-relative rpath so the binary finds the library inside any copy of the repo
gcc -fopenmp -o "$OUT/neutral.hip_dropin" "$OUT/main.o" "$OUT/neutral_data.o" \
    -L"$ROOT/neutral_amd" -lneutral_hip -Wl,-rpath,'$ORIGIN/../../neutral_amd' -lm
rm -f "$OUT/main.o" "$OUT/neutral_data.o"
echo "built $OUT/neutral.hip_dropin"
