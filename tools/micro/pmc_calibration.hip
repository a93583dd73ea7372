// What the SQ counters bench.py's hardware-only figures are made of COUNT, calibrated on kernels whose
// vector-issue load is known by construction (round 5, verdict item 7):
//   SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE
// Run once plainly (prints each kernel's duration, the instructions it issued by construction and the
// shader clock it saw) and once under `rocprofv3 --pmc ...` (tools/pmc_calibration.sh); the log puts
// the two side by side.
//   full4    4 waves per SIMD, eight independent v_fma_f64 chains per lane: the vector unit never idles
//   full1    1 wave per SIMD, the same code: one wave alone issues back to back too (independent chains)
//   chain1   1 wave per SIMD, ONE dependent chain: the unit idles while a result is on its way
//   half4    4 waves per SIMD, blocks of 32 fma separated by s_sleep: idle by construction
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pmc_calibration.hip -o tools/micro/build/pmc_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 8192;
constexpr int kFmaPerIter = 32;

// (shader-clock ticks and 100-MHz ticks of the wave's life, from lane 0 of wave 0 of block 0)
__device__ __forceinline__ void stamp(unsigned long long* clocks, bool begin) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[begin ? 0 : 2] = __builtin_readcyclecounter();
    clocks[begin ? 1 : 3] = wall_clock64();
  }
}

__global__ __launch_bounds__(256) void independent_kernel(double* out, double seed, unsigned long long* clocks) {
  stamp(clocks, true);
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < kFmaPerIter / 8; ++j) {
      a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  stamp(clocks, false);
}

__global__ __launch_bounds__(256) void chain_kernel(double* out, double seed, unsigned long long* clocks) {
  stamp(clocks, true);
  double a0 = seed + threadIdx.x;
  const double m = 1.0000001, c = 1e-9;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < kFmaPerIter; ++j) a0 = __builtin_fma(a0, m, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
  stamp(clocks, false);
}

__global__ __launch_bounds__(256) void sleepy_kernel(double* out, double seed, unsigned long long* clocks) {
  stamp(clocks, true);
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < kFmaPerIter / 8; ++j) {
      a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
    __builtin_amdgcn_s_sleep(16); /* ~1024 cycles: four waves' 4 x 128 cycles of fma fit twice */
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  stamp(clocks, false);
}

int main() {
  int cus = 256;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  double* out;
  unsigned long long* clocks;
  CHECK(hipMalloc((void**)&out, sizeof(double) * 256 * 4 * cus));
  CHECK(hipMalloc((void**)&clocks, sizeof(unsigned long long) * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct Case { const char* name; int kind; int blocks_per_cu; } cases[] = {
      {"full4", 0, 4}, {"full1", 0, 1}, {"chain1", 1, 1}, {"chain4", 1, 4}, {"half4", 2, 4}};
  for (const Case& k : cases) {
    const int grid = cus * k.blocks_per_cu;
    for (int rep = 0; rep < 2; ++rep) { /* (the first launch of each warms the clocks up) */
      CHECK(hipEventRecord(e0, 0));
      if (k.kind == 0) hipLaunchKernelGGL(independent_kernel, dim3(grid), dim3(256), 0, 0, out, 1.0, clocks);
      if (k.kind == 1) hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), 0, 0, out, 1.0, clocks);
      if (k.kind == 2) hipLaunchKernelGGL(sleepy_kernel, dim3(grid), dim3(256), 0, 0, out, 1.0, clocks);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
    }
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[4];
    CHECK(hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost));
    const double ghz = (double)(h[2] - h[0]) / ((double)(h[3] - h[1]) * 10.0); /* ticks per 10 ns */
    const double waves = (double)grid * 4.0;
    const double fma = waves * kIters * kFmaPerIter; /* wave-level v_fma_f64 by construction */
    printf("%-7s grid %5d (%d waves/SIMD)  %8.3f ms  wave-level fma %.4e  shader clock seen %.3f GHz  "
           "fma x 4 cycles / (time x 1024 SIMDs x that clock) = %.3f\n",
           k.name, grid, k.blocks_per_cu, ms, fma, ghz, fma * 4.0 / (ms * 1e-3 * 1024.0 * ghz * 1e9));
  }
  return 0;
}
