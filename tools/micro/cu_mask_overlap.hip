// Can a memory-bound pass run on a few CUs while a compute-bound kernel keeps the rest?
// Two streams with CU masks (hipExtStreamCreateWithCUMask): a VALU-bound kernel on all but
// 16 CUs, a random-gather copy on the 16.  Times: each alone, both together.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/cu_mask_overlap.hip -o tools/micro/build/cu_mask_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void busy(double* out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      a0 = __builtin_fma(a0, 1.0000001, 1e-9); a1 = __builtin_fma(a1, 1.0000001, 1e-9);
      a2 = __builtin_fma(a2, 1.0000001, 1e-9); a3 = __builtin_fma(a3, 1.0000001, 1e-9);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

struct alignas(16) Rec { double v[10]; };

// the export's access pattern: coalesced index read, random 80-B record gather, coalesced stores
__global__ __launch_bounds__(256) void gather(const Rec* rec, const unsigned* slot, double* o0, double* o1, size_t n) {
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
    const Rec r = rec[slot[k]];
    o0[k] = r.v[0] + r.v[2] + r.v[4] + r.v[6] + r.v[8];
    o1[k] = r.v[1] + r.v[3] + r.v[5] + r.v[7] + r.v[9];
  }
}

static float ms(hipEvent_t a, hipEvent_t b) { float t = 0; (void)hipEventElapsedTime(&t, a, b); return t; }

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int small_cus = 16;
  const int words = (cus + 31) / 32;
  std::vector<uint32_t> big(words, 0), small(words, 0);
  // spread the small set over the chip: every (cus/small_cus)-th CU
  for (int c = 0; c < cus; ++c) {
    const bool s = (c % (cus / small_cus)) == 0;
    (s ? small : big)[c / 32] |= 1u << (c % 32);
  }
  hipStream_t sb, ss, plain;
  CHECK(hipExtStreamCreateWithCUMask(&sb, words, big.data()));
  CHECK(hipExtStreamCreateWithCUMask(&ss, words, small.data()));
  CHECK(hipStreamCreate(&plain));
  const size_t n = 100000000;
  Rec* rec; unsigned* slot; double *o0, *o1, *out;
  CHECK(hipMalloc(&rec, sizeof(Rec) * n)); CHECK(hipMalloc(&slot, 4 * n));
  CHECK(hipMalloc(&o0, 8 * n)); CHECK(hipMalloc(&o1, 8 * n)); CHECK(hipMalloc(&out, 8 * 256 * 4096));
  CHECK(hipMemset(rec, 0, sizeof(Rec) * n));
  { std::vector<unsigned> h(n); uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (unsigned)(x % n); }
    CHECK(hipMemcpy(slot, h.data(), 4 * n, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1, e2, e3;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2)); CHECK(hipEventCreate(&e3));
  const int iters = 60000;
  auto run_busy = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(busy, dim3(blocks), dim3(256), 0, s, out, iters); };
  auto run_gather = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, s, rec, slot, o0, o1, n); };
  // warm up
  run_busy(plain, cus * 8); run_gather(plain, cus * 8); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, plain)); run_busy(plain, cus * 8); CHECK(hipEventRecord(e1, plain)); CHECK(hipDeviceSynchronize());
  printf("busy, all %d CUs:                 %8.2f ms\n", cus, ms(e0, e1));
  CHECK(hipEventRecord(e0, plain)); run_gather(plain, cus * 8); CHECK(hipEventRecord(e1, plain)); CHECK(hipDeviceSynchronize());
  printf("gather, all CUs:                   %8.2f ms\n", ms(e0, e1));
  CHECK(hipEventRecord(e0, sb)); run_busy(sb, cus * 8); CHECK(hipEventRecord(e1, sb)); CHECK(hipDeviceSynchronize());
  printf("busy, masked to %d CUs:           %8.2f ms\n", cus - small_cus, ms(e0, e1));
  for (int per : {8, 16, 32}) {
    CHECK(hipEventRecord(e0, ss)); run_gather(ss, small_cus * per); CHECK(hipEventRecord(e1, ss)); CHECK(hipDeviceSynchronize());
    printf("gather, masked to %d CUs, %2d blocks/CU: %8.2f ms\n", small_cus, per, ms(e0, e1));
  }
  CHECK(hipEventRecord(e0, sb)); CHECK(hipEventRecord(e2, ss));
  run_busy(sb, cus * 8); run_gather(ss, small_cus * 16);
  CHECK(hipEventRecord(e1, sb)); CHECK(hipEventRecord(e3, ss)); CHECK(hipDeviceSynchronize());
  printf("together: busy %8.2f ms, gather %8.2f ms\n", ms(e0, e1), ms(e2, e3));
  // without masks: both on plain streams
  hipStream_t p2; CHECK(hipStreamCreate(&p2));
  CHECK(hipEventRecord(e0, plain)); CHECK(hipEventRecord(e2, p2));
  run_busy(plain, cus * 8); run_gather(p2, cus * 8);
  CHECK(hipEventRecord(e1, plain)); CHECK(hipEventRecord(e3, p2)); CHECK(hipDeviceSynchronize());
  printf("together, no masks: busy %8.2f ms, gather %8.2f ms\n", ms(e0, e1), ms(e2, e3));
  return 0;
}
