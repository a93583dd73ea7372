// Litmus test for handing 80-byte particle records from one workgroup to another INSIDE a launch,
// in place, across CUs and XCDs -- the hand-off the asynchronous tile queue of the stream stage
// (neutral_tiled.hip) and the collision stage's work stealing (neutral_kernels.hip) are built on.
//
// N records live in one array, two per 128-byte line on average (80-byte records), so a line is
// shared by records that different workgroups work on at the same time.  Every record hops H times
// from queue to queue (a queue per wave, destination pseudo-random: 7 of 8 hops change XCD).  A hop:
// claim an entry of the own queue, load the record, CHECK that every 16-byte quad of it carries the
// hop count the entry announces (a stale quad carries an older one, a torn record mixes them),
// rewrite it with hop + 1, publish it in the destination's queue.  The protocol variants:
//   0  plain stores, plain loads, no fence            (control: must FAIL if the test can see anything)
//   1  sc1 stores, drained (s_waitcnt vmcnt(0)), then the entry; sc1 loads       (what the library uses)
//   2  plain stores, release fence (agent), entry; acquire fence (agent), plain loads   (the textbook form)
//   3  sc1 stores, drained, entry; acquire fence (agent), plain loads
//   4  as 1, but a record's LAST store (nobody picks it up again in this launch) is a plain store.
//      Records make up to 7 hops fewer than their neighbours in every variant, so records end
//      while the records next to them -- in the same 128-byte line -- still hop.  Suspected of
//      serving stale neighbours from the dirty line it leaves in the storing XCD's L2 while a
//      defect of the stream kernel was hunted; measured: no stale read in 1e7 hops.  (The defect
//      was an inline-asm store whose data register the compiler's next instruction overwrote:
//      a dwordx4 store reads its data up to two wait states after issue -- hence the s_nop 1.)
// Prints per variant: hops, stale / torn records seen, time.  Exit code 1 if variant 1, 2, 3 or 4 saw any.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/handoff_litmus.hip -o tools/micro/build/handoff_litmus
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned kEmpty = 0xFFFFFFFFu;
constexpr int kHopBits = 6;
constexpr unsigned kSlotMask = (1u << (32 - kHopBits)) - 1u;

struct Ctl {
  unsigned live;       // records that have not made their last hop
  unsigned stale;      // quads with an older hop count than announced
  unsigned torn;       // records whose quads disagree / checksum wrong
  unsigned overflow;   // a queue ran out of room (test sized wrongly)
  unsigned timeout;    // waves that gave up waiting
  unsigned hops;
};

__device__ __forceinline__ v4u make_quad(unsigned slot, unsigned hop, unsigned q) {
  v4u v;
  v.x = slot;
  v.y = hop;
  v.z = q;
  v.w = slot ^ (hop * 0x9E3779B9u) ^ q ^ 0xA5A5A5A5u;
  return v;
}

__device__ __forceinline__ void load_plain(const v4u* p, v4u (&q)[5]) {
#pragma unroll
  for (int k = 0; k < 5; ++k) q[k] = p[k];
}
__device__ __forceinline__ void load_sc1(const v4u* p, v4u (&q)[5]) {
  asm volatile(
      "global_load_dwordx4 %0, %5, off sc1\n"
      "global_load_dwordx4 %1, %5, off offset:16 sc1\n"
      "global_load_dwordx4 %2, %5, off offset:32 sc1\n"
      "global_load_dwordx4 %3, %5, off offset:48 sc1\n"
      "global_load_dwordx4 %4, %5, off offset:64 sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4])
      : "v"(p)
      : "memory");
}
__device__ __forceinline__ void store_plain(v4u* p, const v4u (&q)[5]) {
#pragma unroll
  for (int k = 0; k < 5; ++k) p[k] = q[k];
}
__device__ __forceinline__ void store_sc1(v4u* p, const v4u (&q)[5]) {
  asm volatile(
      "global_store_dwordx4 %0, %1, off sc1\n"
      "global_store_dwordx4 %0, %2, off offset:16 sc1\n"
      "global_store_dwordx4 %0, %3, off offset:32 sc1\n"
      "global_store_dwordx4 %0, %4, off offset:48 sc1\n"
      "global_store_dwordx4 %0, %5, off offset:64 sc1\n"
      "s_nop 1\n" /* (the store reads its data registers up to two wait states after issue) */
      :
      : "v"(p), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4])
      : "memory");
}

template <int kVariant>
__global__ __launch_bounds__(64) void hop_kernel(v4u* rec, unsigned* log, unsigned* tail, unsigned* head,
                                                 int nqueues, unsigned cap, unsigned nhops, Ctl* ctl) {
  const int q = (int)blockIdx.x;
  const int lane = (int)threadIdx.x;
  unsigned* my_log = log + (size_t)q * cap;
  unsigned idle = 0;
  for (;;) {
    // claim up to 64 entries of the own queue (lane 0)
    unsigned h = 0, m = 0;
    if (lane == 0) {
      for (;;) {
        h = __hip_atomic_load(&head[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned t = __hip_atomic_load(&tail[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (t > cap) ? cap : t;
        m = (t > h) ? t - h : 0u;
        m = (m > 64u) ? 64u : m;
        if (m == 0) break;
        if (atomicCAS(&head[q], h, h + m) == h) break;
      }
    }
    h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
    m = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    if (m == 0) {
      const unsigned live = __hip_atomic_load(&ctl->live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (live == 0) break;
      if (++idle > (1u << 22)) {
        if (lane == 0) atomicAdd(&ctl->timeout, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    idle = 0;
    unsigned done = 0;
    if ((unsigned)lane < m) {
      unsigned e;
      unsigned spins = 0;
      do {
        e = __hip_atomic_load(&my_log[h + (unsigned)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while (e == kEmpty && ++spins < (1u << 24));
      if (e == kEmpty) {
        atomicAdd(&ctl->timeout, 1u);
      } else {
        const unsigned slot = e & kSlotMask;
        const unsigned hop = e >> (32 - kHopBits);
        if (kVariant == 2 || kVariant == 3) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        v4u r[5];
        v4u* p = rec + (size_t)slot * 5;
        if (kVariant == 1 || kVariant == 4) {
          load_sc1(p, r);
        } else {
          load_plain(p, r);
        }
        bool stale = false, torn = false;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const v4u want = make_quad(slot, hop, (unsigned)k);
          if (r[k].x != slot || r[k].z != (unsigned)k || r[k].w != (r[k].x ^ (r[k].y * 0x9E3779B9u) ^ r[k].z ^ 0xA5A5A5A5u)) {
            torn = true;
          } else if (r[k].y != want.y) {
            stale = true;
          }
        }
        if (stale) atomicAdd(&ctl->stale, 1u);
        if (torn) atomicAdd(&ctl->torn, 1u);
        const unsigned next = hop + 1;
        const unsigned my_hops = nhops - (slot & 7u); /* (neighbours end at different times) */
        v4u w[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) w[k] = make_quad(slot, next, (unsigned)k);
        if (kVariant == 1 || kVariant == 3 || (kVariant == 4 && next < my_hops)) {
          store_sc1(p, w);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          store_plain(p, w);
          if (kVariant == 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          }
        }
        if (next >= my_hops) {
          done = 1;
        } else {
          unsigned x = slot * 2654435761u + next * 40503u;
          x ^= x >> 15;
          x *= 2246822519u;
          x ^= x >> 13;
          const unsigned dest = x % (unsigned)nqueues;
          const unsigned pos = atomicAdd(&tail[dest], 1u);
          if (pos >= cap) {
            atomicAdd(&ctl->overflow, 1u);
            done = 1;
          } else {
            __hip_atomic_store(&log[(size_t)dest * cap + pos], slot | (next << (32 - kHopBits)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    const unsigned ndone = (unsigned)__popcll(__ballot(done != 0));
    if (lane == 0) {
      atomicAdd(&ctl->hops, m);
      if (ndone) atomicSub(&ctl->live, ndone);
    }
  }
}

__global__ void init_kernel(v4u* rec, unsigned* log, unsigned* tail, unsigned* head, int n, int nqueues, unsigned cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nqueues) {
    head[i] = 0;
  }
  if (i < n) {
#pragma unroll
    for (int k = 0; k < 5; ++k) rec[(size_t)i * 5 + k] = make_quad((unsigned)i, 0u, (unsigned)k);
    const int q = i % nqueues;
    const unsigned pos = atomicAdd(&tail[q], 1u);
    log[(size_t)q * cap + pos] = (unsigned)i; /* hop 0 */
  }
}

int main(int argc, char** argv) {
  const int n = (argc > 1) ? atoi(argv[1]) : (1 << 18);
  const unsigned nhops = (argc > 2) ? (unsigned)atoi(argv[2]) : 24u;
  const int nqueues = (argc > 3) ? atoi(argv[3]) : 2048;
  if (nhops >= (1u << kHopBits) || nhops < 9 || (unsigned)n > kSlotMask) {
    printf("bad arguments\n");
    return 2;
  }
  const unsigned cap = (unsigned)((long long)n * nhops / nqueues * 2 + 4096);
  v4u* rec;
  unsigned *log, *tail, *head;
  Ctl* ctl;
  CHECK(hipMalloc((void**)&rec, (size_t)n * 80));
  CHECK(hipMalloc((void**)&log, (size_t)nqueues * cap * 4));
  CHECK(hipMalloc((void**)&tail, (size_t)nqueues * 4));
  CHECK(hipMalloc((void**)&head, (size_t)nqueues * 4));
  CHECK(hipMalloc((void**)&ctl, sizeof(Ctl)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  int bad = 0;
  const char* names[5] = {"plain stores, plain loads, no fence (control)", "sc1 stores + drain, sc1 loads",
                          "plain stores + release fence, acquire fence + plain loads",
                          "sc1 stores + drain, acquire fence + plain loads",
                          "as 1 but a record's last store is plain"};
  for (int rep = 0; rep < 2; ++rep) {
    for (int variant = 0; variant < 5; ++variant) {
      CHECK(hipMemset(log, 0xFF, (size_t)nqueues * cap * 4));
      CHECK(hipMemset(tail, 0, (size_t)nqueues * 4));
      Ctl h = {};
      h.live = (unsigned)n;
      CHECK(hipMemcpy(ctl, &h, sizeof(h), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(init_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, rec, log, tail, head, n, nqueues, cap);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      switch (variant) {
        case 0: hipLaunchKernelGGL(hop_kernel<0>, dim3(nqueues), dim3(64), 0, 0, rec, log, tail, head, nqueues, cap, nhops, ctl); break;
        case 1: hipLaunchKernelGGL(hop_kernel<1>, dim3(nqueues), dim3(64), 0, 0, rec, log, tail, head, nqueues, cap, nhops, ctl); break;
        case 2: hipLaunchKernelGGL(hop_kernel<2>, dim3(nqueues), dim3(64), 0, 0, rec, log, tail, head, nqueues, cap, nhops, ctl); break;
        case 3: hipLaunchKernelGGL(hop_kernel<3>, dim3(nqueues), dim3(64), 0, 0, rec, log, tail, head, nqueues, cap, nhops, ctl); break;
        default: hipLaunchKernelGGL(hop_kernel<4>, dim3(nqueues), dim3(64), 0, 0, rec, log, tail, head, nqueues, cap, nhops, ctl); break;
      }
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
      printf("variant %d (%s): %u hops of %d records over %d queues in %.3f ms: stale %u torn %u | live left %u overflow %u timeouts %u\n",
             variant, names[variant], h.hops, n, nqueues, ms, h.stale, h.torn, h.live, h.overflow, h.timeout);
      if (variant >= 1 && (h.stale || h.torn || h.live || h.overflow || h.timeout)) bad = 1;
    }
  }
  printf(bad ? "FAILED: a hand-off protocol delivered stale or torn records\n" : "OK: variants 1-4 delivered every record whole and current\n");
  return bad;
}
