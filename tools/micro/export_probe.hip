// What bounds the write-back of the tile-sorted 80-B records to the eleven SoA arrays
// (neutral_tiled.hip: export_records_kernel)?  The same permutation in several forms, timed
// with HIP events; the permutation mimics the pipeline's: records sorted by tile (625 tiles),
// ids random inside and across tiles.  Evidence for DESIGN.md section 4.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/export_probe.hip -o tools/micro/build/export_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct alignas(16) Rec {
  double x, y, ox, oy, e, w, dt, mfp;
  int cx, cy;
  unsigned id;
  int dead;
};
static_assert(sizeof(Rec) == 80, "80-byte records");

struct View {
  double *x, *y, *ox, *oy, *e, *w, *dt, *mfp;
  int *cx, *cy, *dead;
};

// the kernel as shipped: one id per thread
template <int kBlock, bool kLoads, bool kStores>
__global__ __launch_bounds__(kBlock) void by_id(const Rec* rec, const unsigned* slot_of_id, View p, int n,
                                                unsigned skip_from) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const unsigned slot = slot_of_id[k];
  if (slot >= skip_from) return;
  Rec r;
  if (kLoads) {
    r = rec[slot];
  } else {
    r.x = r.y = r.ox = r.oy = r.e = r.w = r.dt = r.mfp = (double)slot;
    r.cx = r.cy = r.dead = (int)slot;
  }
  if (kStores) {
    p.x[k] = r.x; p.y[k] = r.y; p.ox[k] = r.ox; p.oy[k] = r.oy; p.e[k] = r.e; p.w[k] = r.w;
    p.dt[k] = r.dt; p.mfp[k] = r.mfp; p.cx[k] = r.cx; p.cy[k] = r.cy; p.dead[k] = r.dead;
  } else if (r.x == 1.234e300 && r.cy == 77 && r.e + r.w + r.ox + r.oy + r.y + r.dt + r.mfp + r.cx + r.dead == 3.0) {
    p.x[k] = r.x;
  }
}

// non-temporal variants: kNtLoad (record reads), kNtStore (array stores)
template <int kBlock, bool kNtLoad, bool kNtStore>
__global__ __launch_bounds__(kBlock) void by_id_nt(const Rec* rec, const unsigned* slot_of_id, View p, int n) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const unsigned slot = slot_of_id[k];
  union { v4u q[5]; Rec r; } u;
  const v4u* src = (const v4u*)&rec[slot];
#pragma unroll
  for (int j = 0; j < 5; ++j) u.q[j] = kNtLoad ? __builtin_nontemporal_load(src + j) : src[j];
  const Rec r = u.r;
  if (kNtStore) {
    __builtin_nontemporal_store(r.x, &p.x[k]); __builtin_nontemporal_store(r.y, &p.y[k]);
    __builtin_nontemporal_store(r.ox, &p.ox[k]); __builtin_nontemporal_store(r.oy, &p.oy[k]);
    __builtin_nontemporal_store(r.e, &p.e[k]); __builtin_nontemporal_store(r.w, &p.w[k]);
    __builtin_nontemporal_store(r.dt, &p.dt[k]); __builtin_nontemporal_store(r.mfp, &p.mfp[k]);
    __builtin_nontemporal_store(r.cx, &p.cx[k]); __builtin_nontemporal_store(r.cy, &p.cy[k]);
    __builtin_nontemporal_store(r.dead, &p.dead[k]);
  } else {
    p.x[k] = r.x; p.y[k] = r.y; p.ox[k] = r.ox; p.oy[k] = r.oy; p.e[k] = r.e; p.w[k] = r.w;
    p.dt[k] = r.dt; p.mfp[k] = r.mfp; p.cx[k] = r.cx; p.cy[k] = r.cy; p.dead[k] = r.dead;
  }
}

// two kernels: records gathered by id into an id-ordered 80-B buffer, then transposed
template <int kBlock>
__global__ __launch_bounds__(kBlock) void gather_records(const Rec* rec, const unsigned* slot_of_id, Rec* out, int n) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < n) out[k] = rec[slot_of_id[k]];
}

// ... in batches whose id-ordered staging buffer stays in the memory-side cache: a gather
// launch (random reads from HBM, stores absorbed by the cache) and a transpose launch (reads
// from the cache, the eleven arrays streamed out) per batch -- reads and writes of HBM
// separated in time
template <int kBlock>
__global__ __launch_bounds__(kBlock) void gather_range(const Rec* rec, const unsigned* slot_of_id, Rec* stage,
                                                       int first, int count) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < count) stage[k] = rec[slot_of_id[first + k]];
}
template <int kBlock>
__global__ __launch_bounds__(kBlock) void transpose_range(const Rec* stage, View p, int first, int count) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= count) return;
  const Rec r = stage[j];
  const int k = first + j;
  p.x[k] = r.x; p.y[k] = r.y; p.ox[k] = r.ox; p.oy[k] = r.oy; p.e[k] = r.e; p.w[k] = r.w;
  p.dt[k] = r.dt; p.mfp[k] = r.mfp; p.cx[k] = r.cx; p.cy[k] = r.cy; p.dead[k] = r.dead;
}

// kPer ids per thread, all slot loads first, then all record loads, then the stores
template <int kBlock, int kPer>
__global__ __launch_bounds__(kBlock) void by_id_ilp(const Rec* rec, const unsigned* slot_of_id, View p, int n) {
  const int base = (blockIdx.x * kBlock) * kPer + threadIdx.x;
  unsigned slot[kPer];
  Rec r[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int k = base + j * kBlock;
    slot[j] = slot_of_id[k < n ? k : n - 1];
  }
#pragma unroll
  for (int j = 0; j < kPer; ++j) r[j] = rec[slot[j]];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int k = base + j * kBlock;
    if (k < n) {
      p.x[k] = r[j].x; p.y[k] = r[j].y; p.ox[k] = r[j].ox; p.oy[k] = r[j].oy; p.e[k] = r[j].e;
      p.w[k] = r[j].w; p.dt[k] = r[j].dt; p.mfp[k] = r[j].mfp; p.cx[k] = r[j].cx; p.cy[k] = r[j].cy;
      p.dead[k] = r[j].dead;
    }
  }
}

// the other direction: records read in slot order (coalesced), eleven scattered stores
template <int kBlock>
__global__ __launch_bounds__(kBlock) void by_slot(const Rec* rec, View p, int n) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const Rec r = rec[s];
  const unsigned k = r.id;
  p.x[k] = r.x; p.y[k] = r.y; p.ox[k] = r.ox; p.oy[k] = r.oy; p.e[k] = r.e; p.w[k] = r.w;
  p.dt[k] = r.dt; p.mfp[k] = r.mfp; p.cx[k] = r.cx; p.cy[k] = r.cy; p.dead[k] = r.dead;
}

// plain copies for scale: n x 80 B read in order + the eleven arrays written
template <int kBlock>
__global__ __launch_bounds__(kBlock) void in_order(const Rec* rec, View p, int n) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const Rec r = rec[k];
  p.x[k] = r.x; p.y[k] = r.y; p.ox[k] = r.ox; p.oy[k] = r.oy; p.e[k] = r.e; p.w[k] = r.w;
  p.dt[k] = r.dt; p.mfp[k] = r.mfp; p.cx[k] = r.cx; p.cy[k] = r.cy; p.dead[k] = r.dead;
}

__global__ void fill_records(Rec* rec, const unsigned* ids, int n) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) {
    Rec r;
    r.x = r.y = r.ox = r.oy = r.e = r.w = r.dt = r.mfp = (double)s;
    r.cx = r.cy = s;
    r.id = ids[s];
    r.dead = 0;
    rec[s] = r;
  }
}

template <typename F>
static float timed(F launch, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 100000000;
  const int ntiles = argc > 2 ? atoi(argv[2]) : 625;
  printf("n = %d particles, %d tiles\n", n, ntiles);
  // permutation: particle id -> tile at random; slots in tile order, random inside a tile
  std::vector<unsigned> slot_of_id(n), id_of_slot(n);
  {
    std::mt19937_64 rng(12345);
    std::vector<unsigned> tile(n);
    std::vector<unsigned> count(ntiles + 1, 0);
    for (int i = 0; i < n; ++i) {
      tile[i] = (unsigned)(rng() % (uint64_t)ntiles);
      count[tile[i] + 1]++;
    }
    for (int t = 0; t < ntiles; ++t) count[t + 1] += count[t];
    std::vector<unsigned> cursor(count.begin(), count.end() - 1);
    // ids visit in a scrambled order so that slots inside a tile are not id-sorted
    std::vector<unsigned> order(n);
    for (int i = 0; i < n; ++i) order[i] = (unsigned)i;
    std::shuffle(order.begin(), order.end(), rng);
    for (int i = 0; i < n; ++i) {
      const unsigned id = order[i];
      const unsigned s = cursor[tile[id]]++;
      slot_of_id[id] = s;
      id_of_slot[s] = id;
    }
  }
  Rec* rec;
  unsigned* d_slot;
  View p;
  CHECK(hipMalloc((void**)&rec, sizeof(Rec) * (size_t)n));
  CHECK(hipMalloc((void**)&d_slot, sizeof(unsigned) * (size_t)n));
  double** f64[] = {&p.x, &p.y, &p.ox, &p.oy, &p.e, &p.w, &p.dt, &p.mfp};
  for (double** f : f64) CHECK(hipMalloc((void**)f, sizeof(double) * (size_t)n));
  int** i32[] = {&p.cx, &p.cy, &p.dead};
  for (int** f : i32) CHECK(hipMalloc((void**)f, sizeof(int) * (size_t)n));
  CHECK(hipMemcpy(d_slot, slot_of_id.data(), sizeof(unsigned) * (size_t)n, hipMemcpyHostToDevice));
  {
    unsigned* d_ids;
    CHECK(hipMalloc((void**)&d_ids, sizeof(unsigned) * (size_t)n));
    CHECK(hipMemcpy(d_ids, id_of_slot.data(), sizeof(unsigned) * (size_t)n, hipMemcpyHostToDevice));
    fill_records<<<(n + 255) / 256, 256>>>(rec, d_ids, n);
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(d_ids));
  }
  const int reps = 5;
  const double gb_min = (double)n * (80.0 + 76.0 + 4.0) / 1e9;
  auto report = [&](const char* name, float ms) {
    printf("%-58s %8.3f ms  (%.2f TB/s of the %.1f GB a permutation must move)\n", name, ms, gb_min / ms, gb_min);
  };
  const unsigned all = ~0u;
  report("in order (no permutation): 256 threads", timed([&] { in_order<256><<<(n + 255) / 256, 256>>>(rec, p, n); }, reps));
  report("by id, as shipped: 256 threads", timed([&] { by_id<256, true, true><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n, all); }, reps));
  report("by id: 1024 threads", timed([&] { by_id<1024, true, true><<<(n + 1023) / 1024, 1024>>>(rec, d_slot, p, n, all); }, reps));
  report("by id: 64 threads", timed([&] { by_id<64, true, true><<<(n + 63) / 64, 64>>>(rec, d_slot, p, n, all); }, reps));
  report("by id, record reads only", timed([&] { by_id<256, true, false><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n, all); }, reps));
  report("by id, stores only", timed([&] { by_id<256, false, true><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n, all); }, reps));
  report("by id, neither (slot_of_id read only)", timed([&] { by_id<256, false, false><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n, all); }, reps));
  report("by id, slots >= 77 % of n skipped", timed([&] { by_id<256, true, true><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n, (unsigned)(0.77 * n)); }, reps));
  report("by id, 2 ids per thread in flight", timed([&] { by_id_ilp<256, 2><<<(n + 511) / 512, 256>>>(rec, d_slot, p, n); }, reps));
  report("by id, 4 ids per thread in flight", timed([&] { by_id_ilp<256, 4><<<(n + 1023) / 1024, 256>>>(rec, d_slot, p, n); }, reps));
  report("by id, non-temporal stores", timed([&] { by_id_nt<256, false, true><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n); }, reps));
  report("by id, non-temporal loads", timed([&] { by_id_nt<256, true, false><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n); }, reps));
  report("by id, non-temporal loads and stores", timed([&] { by_id_nt<256, true, true><<<(n + 255) / 256, 256>>>(rec, d_slot, p, n); }, reps));
  {
    Rec* tmp;
    CHECK(hipMalloc((void**)&tmp, sizeof(Rec) * (size_t)n));
    report("gather into id order only (80-B stores)", timed([&] { gather_records<256><<<(n + 255) / 256, 256>>>(rec, d_slot, tmp, n); }, reps));
    CHECK(hipFree(tmp));
  }
  for (int batch : {400000, 1600000, 3200000, 12800000}) {
    Rec* stage;
    CHECK(hipMalloc((void**)&stage, sizeof(Rec) * (size_t)batch));
    char name[96];
    snprintf(name, sizeof(name), "two launches per batch of %d ids (%.0f MB staged)", batch, batch * 80.0 / 1e6);
    report(name, timed([&] {
      for (int first = 0; first < n; first += batch) {
        const int count = (n - first < batch) ? n - first : batch;
        gather_range<256><<<(count + 255) / 256, 256>>>(rec, d_slot, stage, first, count);
        transpose_range<256><<<(count + 255) / 256, 256>>>(stage, p, first, count);
      }
    }, reps));
    CHECK(hipFree(stage));
  }
  report("by slot (coalesced reads, eleven scattered stores)", timed([&] { by_slot<256><<<(n + 255) / 256, 256>>>(rec, p, n); }, reps));
  return 0;
}
