/*
 * neutral_wave.h -- wave64 helpers shared by the kernel files.
 */
#ifndef NEUTRAL_AMD_WAVE_H
#define NEUTRAL_AMD_WAVE_H

#include <hip/hip_runtime.h>

#include "neutral_kernels.h"

namespace neutral {

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v += __shfl_down(v, off, 64);
  }
  return v; /* valid in lane 0 */
}

__device__ __forceinline__ int lane_rank(unsigned long long mask) {
  /* number of set bits of mask below this lane */
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

/* one atomic per wave and counter (the cuda analog's block tree reduction +
 * host finish, cuda/neutral.k:475-493, collapsed into wave reductions) */
__device__ __forceinline__ void flush_counters(const SolveArgs& a, unsigned nprocessed,
                                               unsigned nfacets, unsigned ncollisions,
                                               unsigned ncensus) {
  const unsigned wf = wave_sum_u32(nfacets);
  const unsigned wc = wave_sum_u32(ncollisions);
  const unsigned wp = wave_sum_u32(nprocessed);
  const unsigned wz = wave_sum_u32(ncensus);
  if ((threadIdx.x & 63) == 0) {
    if (wp) atomicAdd(&a.counters->nprocessed, (unsigned long long)wp);
    if (wf) atomicAdd(&a.counters->nfacets, (unsigned long long)wf);
    if (wc) atomicAdd(&a.counters->ncollisions, (unsigned long long)wc);
    if (wz) atomicAdd(&a.counters->ncensus, (unsigned long long)wz);
  }
}

/* One wave of a launch (wave 0 of workgroup 0) measures the shader clock over its own life:
 * clock_stamp_begin() at the kernel's entry parks the two counters in the launch's StepCounters
 * record (nothing is held in registers), clock_stamp_end() before it leaves turns them into
 * the two tick counts of StepCounters (added up over the launches of a step). */
__device__ __forceinline__ bool clock_wave() { return blockIdx.x == 0 && (threadIdx.x >> 6) == 0; }
__device__ __forceinline__ void clock_stamp_begin(StepCounters* c) {
  if (clock_wave() && (threadIdx.x & 63) == 0) {
    c->clock_parked[0] = __builtin_readcyclecounter();
    c->clock_parked[1] = wall_clock64();
  }
}
__device__ __forceinline__ void clock_stamp_end(StepCounters* c) {
  if (clock_wave() && (threadIdx.x & 63) == 0) {
    const unsigned long long shader = __builtin_readcyclecounter() - c->clock_parked[0];
    const unsigned long long wall = wall_clock64() - c->clock_parked[1];
    atomicAdd(&c->clock_shader_ticks, shader);
    atomicAdd(&c->clock_100mhz_ticks, wall);
  }
}

}  // namespace neutral
#endif
