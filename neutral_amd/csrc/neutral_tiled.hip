/*
 * neutral_tiled.hip -- K3: tile-sorted streaming with the local mesh tile of
 * the tally in LDS, followed by the event-regrouped collision kernel (K2).
 *
 * Why (profiles/r01/ablate_tally.log): once lane divergence is gone, every
 * facet-heavy deck is bound by the memory-side f64 atomic request rate
 * (~1.3-2.1e10 scattered atomics/s for the whole chip), e.g. 86 % of the stream
 * deck's time and all of csp's.  Atomics into LDS do not have that ceiling, but
 * a 400^2 f64 tally (1.28 MB) does not fit 160 KB of LDS -- a window of it does,
 * if the particles a workgroup works on are spatially close.  So, per timestep:
 *
 *   1. tile_count / tile_scan / tile_scatter   counting sort of the LIVE
 *      particle ids by the 16x16-cell tile they start the step in (dead
 *      particles drop out of the work list here instead of being re-scanned
 *      every step, omp3/neutral.c:91-93);
 *   2. stream_kernel   1024-thread persistent workgroups take chunks of one
 *      tile's particles.  A 128x128-cell f64 window of the tally centred on the
 *      tile lives in LDS (128 KB): facet and census tallies inside it are
 *      ds_add_f64, cells outside it fall back to global atomics, and the window
 *      is flushed to the mesh (coalesced rows) when the workgroup moves to
 *      another tile.  Lanes stream their particle (prologue, facets, census)
 *      and are refilled from the chunk like in K2.  A particle whose next event
 *      is a collision is SUSPENDED: its record is stored and its id appended to
 *      the collision queue;
 *   3. history_regroup_kernel (K2) finishes the queued histories with dense
 *      collision waves (neutral_history.h: resume()).
 *
 * Every particle executes the same event bodies with the same RNG counters as
 * in K1, so particle end states are bit-identical; only the summation order of
 * the tally changes.
 */
#include "neutral_kernels.h"

#include "neutral_device.h"
#include "neutral_history.h"
#include "neutral_wave.h"

namespace neutral {

constexpr int kTile = 16;                          /* cells per tile edge */
constexpr int kWindow = 128;                       /* cells per LDS window edge */
constexpr int kMargin = (kWindow - kTile) / 2;     /* window reach beyond the tile */
constexpr int kStreamBlock = 1024;                 /* 16 waves share one window */
#ifndef NEUTRAL_CHUNK_PARTICLES
#define NEUTRAL_CHUNK_PARTICLES 32768
#endif
constexpr int kChunkParticles = NEUTRAL_CHUNK_PARTICLES;
#ifndef NEUTRAL_STREAM_REFILL_MIN
#define NEUTRAL_STREAM_REFILL_MIN 8
#endif
constexpr int kStreamRefillMin = NEUTRAL_STREAM_REFILL_MIN;
constexpr int kSortBlock = 256;

enum Ctrl : int { kCtrlChunkHead = 0, kCtrlNumChunks = 1, kCtrlCollideCount = 2, kCtrlLive = 3 };

__device__ __forceinline__ int tile_of(const TiledArgs& t, int cellx, int celly) {
  return (celly / kTile) * t.tiles_x + (cellx / kTile);
}

/* ---- 1. counting sort of live particle ids by tile ----------------------------- */

/* Adds one per lane to counter[key] for the lanes in `active`, with one atomic
 * per distinct key in the wave; returns this lane's rank among equal keys and
 * the value the counter had before the wave's add (valid for active lanes). */
__device__ __forceinline__ unsigned wave_aggregated_add(unsigned* counter, int key, bool active,
                                                        unsigned& rank_in_key) {
  unsigned base = 0;
  rank_in_key = 0;
  unsigned long long todo = __ballot(active);
  while (todo) { /* wave-uniform loop: one trip per distinct key */
    const int leader = __ffsll((long long)todo) - 1;
    const int leader_key = __shfl(key, leader, 64);
    const unsigned long long same = __ballot(active && key == leader_key);
    unsigned b = 0;
    if ((int)(threadIdx.x & 63) == leader) {
      b = atomicAdd(&counter[leader_key], (unsigned)__popcll(same));
    }
    b = __shfl(b, leader, 64);
    if (active && key == leader_key) {
      base = b;
      rank_in_key = (unsigned)lane_rank(same);
    }
    todo &= ~same;
  }
  return base;
}

__global__ __launch_bounds__(kSortBlock) void tile_count_kernel(SolveArgs a, TiledArgs t) {
  const int stride = gridDim.x * kSortBlock;
  for (int base = blockIdx.x * kSortBlock; base < a.nparticles; base += stride) {
    const int pid = base + threadIdx.x;
    const bool live = pid < a.nparticles && !a.p.dead[pid];
    const int key = live ? tile_of(t, a.p.cellx[pid] - a.x_off, a.p.celly[pid] - a.y_off) : 0;
    unsigned rank;
    wave_aggregated_add(t.tile_count, key, live, rank);
  }
}

/* single workgroup: exclusive scan of the tile counts, then the chunk list */
__global__ __launch_bounds__(1024) void tile_scan_kernel(TiledArgs t) {
  __shared__ unsigned s_part[1024];
  __shared__ unsigned s_chunks[1024];
  const int tid = threadIdx.x;
  const int per = (t.ntiles + 1023) / 1024;
  const int lo = tid * per;
  const int hi = (lo + per < t.ntiles) ? lo + per : t.ntiles;

  unsigned sum = 0;
  unsigned nch = 0;
  for (int i = lo; i < hi; ++i) {
    const unsigned c = t.tile_count[i];
    sum += c;
    nch += (c + kChunkParticles - 1) / kChunkParticles;
  }
  s_part[tid] = sum;
  s_chunks[tid] = nch;
  __syncthreads();
  /* Hillis-Steele inclusive scan over the 1024 partials */
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned a0 = (tid >= off) ? s_part[tid - off] : 0;
    const unsigned c0 = (tid >= off) ? s_chunks[tid - off] : 0;
    __syncthreads();
    s_part[tid] += a0;
    s_chunks[tid] += c0;
    __syncthreads();
  }
  unsigned offset = s_part[tid] - sum;   /* exclusive */
  unsigned chunk = s_chunks[tid] - nch;
  for (int i = lo; i < hi; ++i) {
    const unsigned c = t.tile_count[i];
    t.tile_offset[i] = offset;
    t.tile_cursor[i] = 0;
    for (unsigned b = 0; b < c; b += kChunkParticles) {
      const unsigned e = (b + kChunkParticles < c) ? b + kChunkParticles : c;
      if (chunk < (unsigned)t.max_chunks) {
        t.chunks[chunk] = make_uint4(offset + b, offset + e, (unsigned)i, 0u);
      }
      chunk++;
    }
    offset += c;
  }
  if (tid == 1023) {
    t.ctrl[kCtrlNumChunks] = s_chunks[1023];
    t.ctrl[kCtrlLive] = s_part[1023];
    t.ctrl[kCtrlChunkHead] = 0;
    t.ctrl[kCtrlCollideCount] = 0;
  }
}

__global__ __launch_bounds__(kSortBlock) void tile_scatter_kernel(SolveArgs a, TiledArgs t) {
  const int stride = gridDim.x * kSortBlock;
  for (int base = blockIdx.x * kSortBlock; base < a.nparticles; base += stride) {
    const int pid = base + threadIdx.x;
    const bool live = pid < a.nparticles && !a.p.dead[pid];
    const int key = live ? tile_of(t, a.p.cellx[pid] - a.x_off, a.p.celly[pid] - a.y_off) : 0;
    unsigned rank;
    const unsigned b = wave_aggregated_add(t.tile_cursor, key, live, rank);
    if (live) {
      t.order[t.tile_offset[key] + b + rank] = (unsigned)pid;
    }
  }
}

/* ---- 2. streaming kernel with the LDS tally window ------------------------------ */

__device__ __forceinline__ void flush_window(const SolveArgs& a, double* window, int ox, int oy) {
  /* row-contiguous: one wave instruction adds 64 consecutive cells (512 B) */
  for (int i = threadIdx.x; i < kWindow * kWindow; i += kStreamBlock) {
    const double v = window[i];
    window[i] = 0.0;
    if (v != 0.0) {
      const int gx = ox + (i % kWindow);
      const int gy = oy + (i / kWindow);
      if (gx >= 0 && gx < a.nx && gy >= 0 && gy < a.ny) {
        unsafeAtomicAdd(&a.tally[gy * a.nx + gx], v);
      }
    }
  }
}

template <bool kSameTables>
__global__ __launch_bounds__(kStreamBlock) void stream_kernel(SolveArgs a, TiledArgs t) {
  extern __shared__ double lds_raw[];
  double* window = lds_raw;                                             /* kWindow^2 f64 */
  unsigned short* lds_index = (unsigned short*)(lds_raw + kWindow * kWindow);
  __shared__ int s_chunk;
  __shared__ int s_end;
  __shared__ int s_tile;
  __shared__ int s_cursor;

  /* stage the cs index(es), zero the window */
  CsLookup<const unsigned short*> ix{nullptr, nullptr};
  {
    int used = 0;
    if (a.scatter_index) {
      for (int i = threadIdx.x; i <= a.scatter_index_n; i += kStreamBlock) {
        lds_index[i] = a.scatter_index[i];
      }
      ix.scatter_index = lds_index;
      used = a.scatter_index_n + 1;
    }
    if (!kSameTables && a.absorb_index) {
      for (int i = threadIdx.x; i <= a.absorb_index_n; i += kStreamBlock) {
        lds_index[used + i] = a.absorb_index[i];
      }
      ix.absorb_index = lds_index + used;
    }
    for (int i = threadIdx.x; i < kWindow * kWindow; i += kStreamBlock) {
      window[i] = 0.0;
    }
  }

  const int nchunks = (int)t.ctrl[kCtrlNumChunks];
  int cur_tile = -1;
  WindowTally<kWindow> tally{(lds_double*)window, 0, 0};

  unsigned nfacets = 0;
  unsigned nprocessed = 0;
  unsigned ncensus = 0;

  History h;
  h.ev = kEvEnd;
  int pid = -1;

  for (;;) {
    __syncthreads(); /* the previous chunk is complete (also orders the staging above) */
    if (threadIdx.x == 0) {
      const int c = (int)atomicAdd(&t.ctrl[kCtrlChunkHead], 1u);
      s_chunk = c;
      if (c < nchunks) {
        const uint4 ch = t.chunks[c];
        s_cursor = (int)ch.x;
        s_end = (int)ch.y;
        s_tile = (int)ch.z;
      }
    }
    __syncthreads();
    if (s_chunk >= nchunks) {
      break;
    }
    const int chunk_end = s_end;
    if (s_tile != cur_tile) {
      /* move the window: flush what the previous tile accumulated */
      if (cur_tile >= 0) {
        flush_window(a, window, tally.ox, tally.oy);
      }
      cur_tile = s_tile;
      tally.ox = (cur_tile % t.tiles_x) * kTile - kMargin;
      tally.oy = (cur_tile / t.tiles_x) * kTile - kMargin;
      __syncthreads();
    }

    /* ---- this wave's share of the chunk: refill / stream passes ---- */
    bool has = false;      /* lane holds a particle that wants a STREAM pass */
    bool drained = false;
    for (;;) {
      const unsigned long long m_empty = __ballot(!has);
      const int n_empty = drained ? 0 : __popcll(m_empty);
      const int n_stream = 64 - __popcll(m_empty);
      if (n_empty + n_stream == 0) {
        break;
      }
      bool suspend = false;
      if (n_empty >= kStreamRefillMin || n_stream == 0) {
        /* REFILL: take n_empty ids of the chunk */
        int base = 0;
        if ((threadIdx.x & 63) == 0) {
          base = atomicAdd(&s_cursor, n_empty);
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= chunk_end) {
          drained = true;
        } else {
          const int mine = base + lane_rank(m_empty);
          if (!has && mine < chunk_end) {
            pid = (int)t.order[mine];
            nprocessed++;
            load_particle(h, a, pid);
            prologue<kSameTables>(h, a, ix);
            decide(h, a);
            has = true;
            suspend = (h.ev == kEvCollision);
          }
        }
      } else if (has) {
        /* STREAM: one facet crossing, or the end of the history */
        if (h.ev == kEvFacet) {
          nfacets++;
          cross_facet(h, a, tally);
          decide(h, a);
          suspend = (h.ev == kEvCollision);
        } else {
          if (h.ev == kEvCensus) {
            ncensus++;
            census(h, a, tally);
          }
          store_particle(h, a, pid); /* kEvEnd: the loop at omp3/neutral.c:134 exits */
          has = false;
        }
      }
      /* suspended histories: store the record, queue the id (one atomic per wave) */
      const unsigned long long m_susp = __ballot(suspend);
      if (m_susp) {
        unsigned qbase = 0;
        const int leader = __ffsll((long long)m_susp) - 1;
        if ((int)(threadIdx.x & 63) == leader) {
          qbase = atomicAdd(&t.ctrl[kCtrlCollideCount], (unsigned)__popcll(m_susp));
        }
        qbase = __shfl(qbase, leader, 64);
        if (suspend) {
          store_particle(h, a, pid);
          t.collide_queue[qbase + lane_rank(m_susp)] = (unsigned)pid;
          has = false;
        }
      }
    }
  }
  if (cur_tile >= 0) {
    flush_window(a, window, tally.ox, tally.oy);
  }
  flush_counters(a, nprocessed, nfacets, 0u, ncensus);
}

/* ---- launcher ---------------------------------------------------------------------- */

size_t tiled_lds_bytes(const SolveArgs& a) {
  size_t lds = sizeof(double) * kWindow * kWindow;
  if (a.scatter_index) {
    lds += sizeof(unsigned short) * (a.scatter_index_n + 1);
  }
  if (!a.same_tables && a.absorb_index) {
    lds += sizeof(unsigned short) * (a.absorb_index_n + 1);
  }
  return (lds + 15) & ~(size_t)15;
}

void tiled_geometry(int nx, int ny, int nparticles, int* tiles_x, int* tiles_y, int* max_chunks) {
  *tiles_x = (nx + kTile - 1) / kTile;
  *tiles_y = (ny + kTile - 1) / kTile;
  /* every tile can end with one partial chunk */
  *max_chunks = (*tiles_x) * (*tiles_y) + nparticles / kChunkParticles + 1;
}

hipError_t launch_solve_tiled(const SolveArgs& a, const TiledArgs& t, hipStream_t stream) {
  if (a.nparticles <= 0) {
    return hipSuccess;
  }
  hipError_t err = hipMemsetAsync(t.tile_count, 0, sizeof(unsigned) * t.ntiles, stream);
  if (err != hipSuccess) {
    return err;
  }
  int sort_grid = (a.nparticles + kSortBlock - 1) / kSortBlock;
  if (sort_grid > 8192) {
    sort_grid = 8192;
  }
  hipLaunchKernelGGL(tile_count_kernel, dim3(sort_grid), dim3(kSortBlock), 0, stream, a, t);
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, t);
  hipLaunchKernelGGL(tile_scatter_kernel, dim3(sort_grid), dim3(kSortBlock), 0, stream, a, t);

  /* one 1024-thread workgroup per CU (the window takes most of the LDS) */
  int dev = 0;
  int cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  const size_t lds = tiled_lds_bytes(a);
  if (a.same_tables) {
    (void)hipFuncSetAttribute((const void*)stream_kernel<true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(stream_kernel<true>, dim3(cus), dim3(kStreamBlock), lds, stream, a, t);
  } else {
    (void)hipFuncSetAttribute((const void*)stream_kernel<false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(stream_kernel<false>, dim3(cus), dim3(kStreamBlock), lds, stream, a, t);
  }
  err = hipGetLastError();
  if (err != hipSuccess) {
    return err;
  }

  /* 3. the suspended histories: K2 over the collision queue */
  SolveArgs c = a;
  c.queue = t.collide_queue;
  c.queue_len = &t.ctrl[kCtrlCollideCount];
  return launch_solve(c, kVariantEventSorted, stream);
}

}  // namespace neutral
