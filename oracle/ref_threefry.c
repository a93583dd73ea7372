/*
 * ref_threefry.c -- TEST INFRASTRUCTURE.  Thin export of the REFERENCE's own
 * Random123 Threefry2x64 (compiled in place from $(REFERENCE)/Random123, never
 * copied) so tests can pin oracle/neutral_oracle.c and the HIP kernels to it
 * bit for bit.  Built only where the reference tree is present; the product
 * lands in oracle/_ref/ (git-ignored, travels to the GPU box).
 *
 * Mirrors the call shape of omp3/neutral.c:636-644.
 */
#include <stdint.h>

#include "Random123/threefry.h"

void ref_threefry2x64(uint64_t c0, uint64_t c1, uint64_t k0, uint64_t k1,
                      uint64_t* out0, uint64_t* out1) {
  threefry2x64_ctr_t ctr;
  threefry2x64_key_t key;
  ctr.v[0] = c0;
  ctr.v[1] = c1;
  key.v[0] = k0;
  key.v[1] = k1;
  threefry2x64_ctr_t r = threefry2x64(ctr, key);
  *out0 = r.v[0];
  *out1 = r.v[1];
}

/* many draws in one call: counters ctr0..ctr0+n-1 for one (k0,k1) */
void ref_threefry2x64_stream(uint64_t ctr0, uint64_t k0, uint64_t k1, int n,
                             uint64_t* out) {
  for (int i = 0; i < n; ++i) {
    ref_threefry2x64(ctr0 + (uint64_t)i, 0, k0, k1, &out[2 * i], &out[2 * i + 1]);
  }
}
