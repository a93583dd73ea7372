// Does the fast policy's scatter_cosine() (neutral_device.h: the second quotient and the second root seeded from the first)
// deliver the bits of the IEEE evaluation of omp3/neutral.c:263-265?  Random energies over the tables' range, random
// scattering samples; counts the cosines, and the parts, that differ.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I neutral_amd/csrc tools/micro/scatter_cosine.hip -o tools/micro/build/scatter_cosine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "neutral_device.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace neutral;

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters, uint64_t seed) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long bad = 0, bad_speed = 0, bad_dir = 0;
  for (int i = 0; i < iters; ++i) {
    const uint64_t h = mix(seed + tid * 0x9E3779B97F4A7C15ull + (uint64_t)i);
    const double u = u64_to_unit(h), v = u64_to_unit(mix(h + 1));
    const double e = 1.0e-2 * exp2(u * 33.2);            /* 1e-2 ... 1e8 eV */
    const double mu_cm = 1.0 - 2.0 * v;
    const double e_new = (e * (kMassNo * kMassNo + 2.0 * kMassNo * mu_cm + 1.0)) / ((kMassNo + 1.0) * (kMassNo + 1.0));
    const double want = 0.5 * ((kMassNo + 1.0) * sqrt(e_new / e) - (kMassNo - 1.0) * sqrt(e / e_new));
    double root_ratio, inv_root_ratio;
    const double got = scatter_cosine<false>(e, e_new, root_ratio, inv_root_ratio);
    bad += (__double_as_longlong(got) != __double_as_longlong(want));
    /* the speed after the scatter, from the speed before it (omp3/neutral.c:297) */
    const double speed = sqrt((2.0 * e * kEvToJ) / kParticleMass);
    const double want_speed = sqrt((2.0 * e_new * kEvToJ) / kParticleMass);
    const double got_speed = speed_after_scatter(e_new, speed, refined_reciprocal(speed), root_ratio, inv_root_ratio);
    bad_speed += (__double_as_longlong(got_speed) != __double_as_longlong(want_speed));
    /* the direction's two reciprocals 1 / (omega speed) off one seed (refresh_direction_plain_or_wrapped) */
    const double ang = 6.283185307179586 * u64_to_unit(mix(h + 2));
    double ox = cos(ang), oy = sin(ang);
    if ((h & 1023) == 0) ox *= exp2(-100.0 * u); /* (some nearly axis-parallel) */
    const double ux = ox * want_speed, uy = oy * want_speed;
    const double r_both = refined_reciprocal(ux * uy);
    const double qx = r_both * uy, qy = r_both * ux;
    const double gx = __builtin_fma(__builtin_fma(-ux, qx, 1.0), qx, qx);
    const double gy = __builtin_fma(__builtin_fma(-uy, qy, 1.0), qy, qy);
    bad_dir += (__double_as_longlong(gx) != __double_as_longlong(1.0 / ux)) + (__double_as_longlong(gy) != __double_as_longlong(1.0 / uy));
  }
  atomicAdd(&out[0], bad);
  atomicAdd(&out[1], bad_speed);
  atomicAdd(&out[2], bad_dir);
}

int main() {
  unsigned long long* d; CHECK(hipMalloc(&d, 64)); CHECK(hipMemset(d, 0, 64));
  const int blocks = 256 * 32, iters = 4096;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, iters, 2026ull);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[8]; CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
  printf("scatters %.3e: %llu cosines, %llu speeds and %llu direction reciprocals differ from the IEEE evaluation\n", (double)blocks * 256 * iters, h[0], h[1], h[2]);
  return 0;
}
