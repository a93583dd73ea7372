// How often do a quotient and a square root built on ONE refinement step of v_rcp_f64 / v_rsq_f64 differ from the
// IEEE results (the compiler's own division and sqrt)?  And how good are the two seeds?  (round 5: DESIGN section 4 item 25)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/one_step.hip -o tools/micro/build/one_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__device__ __forceinline__ double unit(uint64_t u) { return __longlong_as_double((long long)((u >> 12) | 0x3FF0000000000000ull)); } // [1, 2)

__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters, uint64_t seed) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long bad_div1 = 0, bad_div2 = 0, bad_sqrt8 = 0, bad_sqrt10 = 0;
  double worst_rcp = 0.0, worst_rsq = 0.0;
  for (int i = 0; i < iters; ++i) {
    const uint64_t h = mix(seed + tid * 0x9E3779B97F4A7C15ull + (uint64_t)i);
    const double a = unit(h) * 3.0e5, b = unit(mix(h + 1)) * 7.0e-3;
    const double want = a / b;
    double r = __builtin_amdgcn_rcp(b);
    worst_rcp = fmax(worst_rcp, fabs(__builtin_fma(-b, r, 1.0)));
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    { const double q0 = a * r; const double rem = __builtin_fma(-b, q0, a); bad_div1 += (__builtin_fma(rem, r, q0) != want); }
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    { const double q0 = a * r; const double rem = __builtin_fma(-b, q0, a); bad_div2 += (__builtin_fma(rem, r, q0) != want); }
    const double x = unit(mix(h + 2)) * ((h & 1) ? 2.0 : 1.0) * 1.0e3;
    const double root = sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    worst_rsq = fmax(worst_rsq, fabs(__builtin_fma(-x * y, y, 1.0)) * 0.5);
    const double g0 = x * y, h0 = 0.5 * y;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    bad_sqrt8 += (__builtin_fma(d0, h1, g1) != root);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    bad_sqrt10 += (__builtin_fma(d1, h1, g2) != root);
  }
  atomicAdd(&out[0], bad_div1); atomicAdd(&out[1], bad_div2); atomicAdd(&out[2], bad_sqrt8); atomicAdd(&out[3], bad_sqrt10);
  atomicMax(&out[4], (unsigned long long)__double_as_longlong(worst_rcp));
  atomicMax(&out[5], (unsigned long long)__double_as_longlong(worst_rsq));
}

int main() {
  unsigned long long* d; CHECK(hipMalloc(&d, 64)); CHECK(hipMemset(d, 0, 64));
  const int blocks = 256 * 32, iters = 4096;   // 3.4e10 samples
  for (int rep = 0; rep < 1; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, iters, 12345ull + rep);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[8]; CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
  double wr, ws; memcpy(&wr, &h[4], 8); memcpy(&ws, &h[5], 8);
  const double n = (double)blocks * 256 * iters;
  printf("samples %.3e\n quotient, one Newton step on v_rcp_f64 : %llu differ from a / b\n quotient, two steps                   : %llu\n"
         " sqrt, one coupled step + correction (8 operations) : %llu differ from sqrt()\n sqrt, the ten operations              : %llu\n"
         " worst seed error: v_rcp_f64 %.3e (2^%.1f)   v_rsq_f64 %.3e (2^%.1f)\n", n, h[0], h[1], h[2], h[3], wr, log2(wr), ws, log2(ws));
  return 0;
}
