// Scattered f64 atomic adds into a mesh-sized array: one shared copy against one copy per XCD
// (waves pick the copy of the XCD they run on, XCC_ID hardware register).  Evidence for how the
// tally atomics of the un-windowed paths are bounded.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/micro/atomic_rate.hip -o gpurun_out/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// mode 0: one shared array; mode 1: copy per XCD; mode 2: shared array, neighbouring cells per lane
// (a particle track: consecutive adds of a lane go to adjacent cells); mode 3: the flush pattern
// of an LDS tally window -- the 64 lanes of a wave add to 64 CONSECUTIVE cells (one 512-B row
// segment per wave instruction), segments at random rows; mode 4: copy per XCD with atomics of
// WORKGROUP scope (no sc1: may the XCD's own L2 perform them?); mode 5: the same, wavefront scope;
// mode 6: copy per XCD, workgroup scope, 64-cell row segments; mode 7: shared array, eight lanes
// per 64-byte line, the wave's eight lines at random places (what a write-combining flush of
// per-history line buffers would issue)
template <int kMode>
__global__ __launch_bounds__(256) void scatter_add(double* tally, unsigned ncells, int iters, unsigned* xcc_seen) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  double* base = tally;
  if (kMode == 1 || (kMode >= 4 && kMode != 7)) {
    const unsigned x = xcc_id();
    base = tally + (size_t)x * ncells;
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1u << x);
  }
  unsigned cell = hash32(tid) % ncells;
  for (int i = 0; i < iters; ++i) {
    if (kMode == 2) {
      cell = (cell + 1 < ncells) ? cell + 1 : 0;
    } else if (kMode == 3 || kMode == 6) {
      const unsigned wave = tid >> 6;
      const unsigned seg = hash32(wave * 0x9e3779b9u + (unsigned)i) % (ncells / 64);
      cell = seg * 64 + (tid & 63);
    } else if (kMode == 7) {
      const unsigned group = tid >> 3;
      const unsigned line = hash32(group * 0x9e3779b9u + (unsigned)i * 0x85ebca6bu) % (ncells / 8);
      cell = line * 8 + (tid & 7);
    } else {
      cell = hash32(cell + 0x9e3779b9u * (unsigned)i + tid) % ncells;
    }
    if (kMode == 4 || kMode == 6) {
      __hip_atomic_fetch_add(&base[cell], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (kMode == 5) {
      __hip_atomic_fetch_add(&base[cell], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
      unsafeAtomicAdd(&base[cell], 1.0);
    }
  }
}

template <typename K>
static int run(const char* name, K kernel, double* tally, unsigned ncells, int copies, unsigned* d_seen, int blocks, int iters) {
  CHECK(hipMemset(tally, 0, sizeof(double) * (size_t)ncells * copies));
  CHECK(hipMemset(d_seen, 0, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tally, ncells, 8, d_seen);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(tally, 0, sizeof(double) * (size_t)ncells * copies));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tally, ncells, iters, d_seen);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // checksum: every add must have landed
  std::vector<double> h((size_t)ncells * copies);
  CHECK(hipMemcpy(h.data(), tally, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  double sum = 0; for (double v : h) sum += v;
  unsigned seen = 0; CHECK(hipMemcpy(&seen, d_seen, 4, hipMemcpyDeviceToHost));
  const double adds = (double)blocks * 256.0 * iters;
  printf("%-34s cells %9u  %8.3f ms  %8.2f G adds/s  sum %s  xcc mask 0x%x\n", name, ncells, ms, adds / (ms * 1e-3) / 1e9,
         (sum == adds) ? "ok" : "LOST UPDATES", seen);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;
  unsigned* d_seen; CHECK(hipMalloc(&d_seen, 4));
  for (unsigned n : {400u * 400u, 4000u * 4000u}) {
    double* tally; CHECK(hipMalloc(&tally, sizeof(double) * (size_t)n * 8));
    const int iters = 2000;
    if (run("shared array, random cells", scatter_add<0>, tally, n, 1, d_seen, blocks, iters)) return 1;
    if (run("copy per XCD, random cells", scatter_add<1>, tally, n, 8, d_seen, blocks, iters)) return 1;
    if (run("shared array, track of cells", scatter_add<2>, tally, n, 1, d_seen, blocks, iters)) return 1;
    if (run("shared array, 64-cell row segments", scatter_add<3>, tally, n, 1, d_seen, blocks, iters)) return 1;
    if (run("shared array, 8 lanes per random line", scatter_add<7>, tally, n, 1, d_seen, blocks, iters)) return 1;
    if (run("copy per XCD, workgroup scope", scatter_add<4>, tally, n, 8, d_seen, blocks, iters)) return 1;
    if (run("copy per XCD, wavefront scope", scatter_add<5>, tally, n, 8, d_seen, blocks, iters)) return 1;
    if (run("copy per XCD, wg scope, row segments", scatter_add<6>, tally, n, 8, d_seen, blocks, iters)) return 1;
    CHECK(hipFree(tally));
  }
  return 0;
}
