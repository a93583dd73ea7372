// Measured VALU issue rates on the box: the denominator for kernels that are bound
// by fp64 / int64 vector issue rather than by HBM.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_peak.hip -o gpurun_out/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 4096;

__global__ __launch_bounds__(256) void fma64_kernel(double* out, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
      a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// one dependent chain per lane: what a single wave can issue back to back
__global__ __launch_bounds__(256) void fma64_chain_kernel(double* out, double seed) {
  double a0 = seed + threadIdx.x;
  const double m = 1.0000001, c = 1e-9;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) a0 = __builtin_fma(a0, m, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}

__global__ __launch_bounds__(256) void rcp64_kernel(double* out, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

// 64-bit integer add as the compiler emits it for Threefry (v_lshl_add_u64); the xor
// between the adds keeps the chain from folding into a closed form
__global__ __launch_bounds__(256) void add64i_kernel(uint64_t* out, uint64_t seed) {
  uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a0) : "v"(a1));
      asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a1) : "v"(a2));
      asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a2) : "v"(a3));
      asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a3) : "v"(a0));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

#define ASM4(INSN)                                             \
  asm volatile(INSN : "+v"(a0) : "v"(a1), "v"(a2));            \
  asm volatile(INSN : "+v"(a1) : "v"(a2), "v"(a3));            \
  asm volatile(INSN : "+v"(a2) : "v"(a3), "v"(a0));            \
  asm volatile(INSN : "+v"(a3) : "v"(a0), "v"(a1));

#define F64_KERNEL(NAME, INSN)                                                   \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed) {        \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;       \
    _Pragma("unroll 1") for (int i = 0; i < kIters; ++i) {                       \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { ASM4(INSN) }               \
    }                                                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;              \
  }

F64_KERNEL(mul64_kernel, "v_mul_f64 %0, %0, %1")
F64_KERNEL(add64f_kernel, "v_add_f64 %0, %0, %1")
F64_KERNEL(divfixup_kernel, "v_div_fixup_f64 %0, %0, %1, %2")
F64_KERNEL(divfmas_kernel, "v_div_fmas_f64 %0, %0, %1, %2")
F64_KERNEL(ldexp_kernel, "v_ldexp_f64 %0, %0, 1")
F64_KERNEL(rsq64_kernel, "v_rsq_f64 %0, %0")
F64_KERNEL(sqrt64_kernel, "v_sqrt_f64 %0, %0")

__global__ __launch_bounds__(256) void divscale_kernel(double* out, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a0) : "v"(a1) : "vcc");
      asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a1) : "v"(a2) : "vcc");
      asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a2) : "v"(a3) : "vcc");
      asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a3) : "v"(a0) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

__global__ __launch_bounds__(256) void cvt_u32_f64_kernel(double* out, double seed) {
  double a0 = seed, a1 = seed, a2 = seed, a3 = seed;
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a0) : "v"(u0));
      asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a1) : "v"(u1));
      asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a2) : "v"(u2));
      asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a3) : "v"(u3));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

__global__ __launch_bounds__(256) void alignbit_kernel(unsigned* out, unsigned seed) {
  unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = __builtin_amdgcn_alignbit(a0, a1, 7); a1 = __builtin_amdgcn_alignbit(a1, a2, 9);
      a2 = __builtin_amdgcn_alignbit(a2, a3, 11); a3 = __builtin_amdgcn_alignbit(a3, a0, 13);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}


// 32-bit integer VALU of the path (Threefry's xor, carry adds, selects): does the guide's
// "2 cycles per wave64 instruction with >= 2 waves per SIMD" apply to them, or do they hold
// the SIMD for 4 cycles like the f64 / 64-bit ones?
#define U32_KERNEL(NAME, BODY)                                                   \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed) {    \
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;     \
    _Pragma("unroll 1") for (int i = 0; i < kIters; ++i) {                       \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { BODY }                     \
    }                                                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;              \
  }

// byte-granular alternatives to v_alignbit_b32 for Threefry's rotations by 16 and 24
U32_KERNEL(perm32_kernel,
  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "s"(0x01000706u));
  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "s"(0x01000706u));
  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "s"(0x01000706u));
  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(a0), "s"(0x01000706u));)

U32_KERNEL(alignbyte32_kernel,
  asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(a0) : "v"(a1));
  asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(a1) : "v"(a2));
  asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(a2) : "v"(a3));
  asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(a3) : "v"(a0));)

U32_KERNEL(lshlor32_kernel,
  asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(a0) : "v"(a1));
  asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(a1) : "v"(a2));
  asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(a2) : "v"(a3));
  asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(a3) : "v"(a0));)

U32_KERNEL(lshrrev32_kernel,
  asm volatile("v_lshrrev_b32 %0, 7, %1" : "+v"(a0) : "v"(a1));
  asm volatile("v_lshrrev_b32 %0, 7, %1" : "+v"(a1) : "v"(a2));
  asm volatile("v_lshrrev_b32 %0, 7, %1" : "+v"(a2) : "v"(a3));
  asm volatile("v_lshrrev_b32 %0, 7, %1" : "+v"(a3) : "v"(a0));)

U32_KERNEL(xor32_kernel,
  asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(a1));
  asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a1) : "v"(a2));
  asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a2) : "v"(a3));
  asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a3) : "v"(a0));)
U32_KERNEL(addco32_kernel,
  asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a0) : "v"(a1) : "vcc");
  asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a1) : "v"(a2) : "vcc");
  asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a2) : "v"(a3) : "vcc");
  asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a3) : "v"(a0) : "vcc");)
U32_KERNEL(addu32_kernel,
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1));
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(a2));
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(a3));
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(a0));)
U32_KERNEL(cndmask32_kernel,
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1) : "vcc");
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(a2) : "vcc");
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(a3) : "vcc");
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(a0) : "vcc");)
U32_KERNEL(mov32_kernel,
  asm volatile("v_mov_b32 %0, %1" : "+v"(a0) : "v"(a1));
  asm volatile("v_mov_b32 %0, %1" : "+v"(a1) : "v"(a2));
  asm volatile("v_mov_b32 %0, %1" : "+v"(a2) : "v"(a3));
  asm volatile("v_mov_b32 %0, %1" : "+v"(a3) : "v"(a0));)

__global__ __launch_bounds__(256) void fma32_kernel(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "v"(a3));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a0));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a0), "v"(a1));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <typename K, typename T>
static int run(const char* name, K kernel, T* out, T seed, double ops_per_thread, int blocks, const char* unit) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, seed);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, seed);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double ops = ops_per_thread * 256.0 * blocks;
  printf("%-22s %8.3f ms  %10.3f G%s/s  (%d blocks)\n", name, best, ops / (best * 1e-3) / 1e9, unit, blocks);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
  void* buf; CHECK(hipMalloc(&buf, 8ull * 256 * cus * 16));
  for (int per_cu : {4, 8, 12, 16}) {  // 1, 2, 3, 4 waves per SIMD
    const int blocks = cus * per_cu;
    printf("-- %d waves/SIMD\n", per_cu / 4);
    if (run("fma f64 (8 indep.)", fma64_kernel, (double*)buf, 1.0, kIters * 32.0, blocks, "FMA")) return 1;
    if (run("fma f64 (1 chain)", fma64_chain_kernel, (double*)buf, 1.0, kIters * 32.0, blocks, "FMA")) return 1;
    if (run("rcp f64", rcp64_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
    if (run("add u64", add64i_kernel, (uint64_t*)buf, (uint64_t)3, kIters * 32.0, blocks, "op")) return 1;
    if (run("alignbit b32", alignbit_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
    if (run("v_xor_b32", xor32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
    if (per_cu == 12) {
      if (run("v_perm_b32", perm32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_alignbyte_b32", alignbyte32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_lshl_or_b32", lshlor32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_lshrrev_b32", lshrrev32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
    }
    if (run("v_add_co_u32", addco32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
    if (run("v_add_u32", addu32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
    if (run("v_cndmask_b32", cndmask32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
    if (run("v_mov_b32", mov32_kernel, (unsigned*)buf, 3u, kIters * 32.0, blocks, "op")) return 1;
    if (run("v_fma_f32", fma32_kernel, (float*)buf, 1.0f, kIters * 32.0, blocks, "op")) return 1;
    if (per_cu == 16) {
      if (run("v_mul_f64", mul64_kernel, (double*)buf, 1.0, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_add_f64", add64f_kernel, (double*)buf, 1.0, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_div_scale_f64", divscale_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_div_fmas_f64", divfmas_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_div_fixup_f64", divfixup_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_ldexp_f64", ldexp_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_rsq_f64", rsq64_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_sqrt_f64", sqrt64_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
      if (run("v_cvt_f64_u32", cvt_u32_f64_kernel, (double*)buf, 1.5, kIters * 32.0, blocks, "op")) return 1;
    }
  }
  // peak if every SIMD issued one wave64 instruction per 4 cycles at the reported clock
  printf("nominal: %d CUs x 4 SIMD x 16 lanes x %.2f GHz = %.1f Glane-op/s\n", cus, p.clockRate / 1e6,
         cus * 4 * 16 * (p.clockRate / 1e6));
  return 0;
}
