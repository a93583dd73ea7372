/*
 * neutral_tiled.hip -- K3: tile-sorted streaming with the local mesh tile of
 * the tally in LDS, followed by the event-regrouped collision kernel (K2).
 *
 * Why (profiles/r01/ablate_tally.log): once lane divergence is gone, every
 * facet-heavy deck is bound by the memory-side f64 atomic request rate
 * (~2.3e10 scattered atomics/s for the whole chip), e.g. 86 % of the stream
 * deck's time and all of csp's.  Atomics into LDS do not have that ceiling, but
 * a 400^2 f64 tally (1.28 MB) does not fit 160 KB of LDS -- a window of it does,
 * if the particles a workgroup works on are spatially close.  So, per timestep:
 *
 *   0. the variant works on a private array of 80-byte particle RECORDS kept in
 *      tile order from step to step (neutral_kernels.h: ParticleRec); the SoA
 *      store of the interface is imported once and kept current by the kernels
 *      that finish a history (or written back on demand, lazy export);
 *   1. tile_count / tile_scan / tile_scatter / tile_chunks   an own COUNTING SORT
 *      of the LIVE records by the T x T-cell tile they start the step in (one
 *      histogram pass over the 4-byte record summaries, one scan over the tiles,
 *      one placement pass: every workgroup reserves its share of each tile's
 *      range with one atomic per tile it holds; dead particles get the last
 *      bucket and drop out of the work list instead of being re-scanned every
 *      step, omp3/neutral.c:91-93).  T is chosen per problem from the particle
 *      density (16 for the dense BASELINE configurations ... 128 for the
 *      reference's decks as shipped, 4000^2 cells and 1e6 particles);
 *   2. stream_kernel   1024-thread persistent workgroups take chunks of one
 *      tile's particles.  A 128x128-cell f64 window of the tally centred on the
 *      tile lives in LDS (128 KB): facet and census tallies inside it are
 *      ds_add_f64, cells outside it fall back to global atomics, and the window
 *      is flushed to the mesh (coalesced rows) when the workgroup moves to
 *      another tile.  Lanes stream their particle (prologue, facets, census)
 *      and are refilled from the chunk like in K2.  A particle whose next event
 *      is a collision is SUSPENDED: its record is stored and marked for the
 *      collision stage.  A particle that leaves the window with many facets
 *      still ahead (fast particles: the stream deck crosses 553 cells per step)
 *      is handed to the NEXT PASS of steps 1-2, which sorts the migrants by the
 *      tile they have reached and works on their records in place; passes
 *      repeat until nobody migrates.  The host does not wait between passes: it
 *      enqueues as many as the previous timestep needed plus one (a pass without
 *      migrants costs four empty launches) and looks at the migrant counter once,
 *      together with the step's event counters;
 *   3. history_regroup_kernel (K2) finishes the suspended histories with dense
 *      collision waves (neutral_history.h: resume()).
 *
 * Every particle executes the same event bodies with the same RNG counters as
 * in K1, so particle end states are bit-identical; only the summation order of
 * the tally changes.
 */
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "neutral_kernels.h"

#include "neutral_device.h"
#include "neutral_history.h"
#include "neutral_wave.h"

namespace neutral {

/* (cells per LDS window edge: window_cells() in neutral_history.h -- 128; two windows of 88 or 100 with the
 * scalar flux) */
/* (-DNEUTRAL_STREAM_BLOCK=768: the build of the co-residency experiment of round 5 -- three waves
 * per SIMD stream as fast as four, and leave a wave slot per SIMD to another kernel) */
#ifndef NEUTRAL_STREAM_BLOCK
#define NEUTRAL_STREAM_BLOCK 1024
#endif
constexpr int kStreamBlock = NEUTRAL_STREAM_BLOCK; /* 16 waves share one window */
/* A chunk is what one workgroup takes at a time from one tile's particles: big
 * enough to amortise its two barriers and a window move, small enough that
 * every workgroup gets several (the host picks the size from the particle
 * count: tiled_chunk_particles). */
constexpr int kChunkParticlesMax = 32768;
/* (the floor is one particle per lane of a workgroup: where tiles hold about that many per
 * pass -- the reference's decks as shipped, a thousand particles per tile -- a chunk of a
 * few thousand puts a whole tile, four histories per lane one after the other, on ONE
 * workgroup: stream 4000^2 / 1e6 64.9 -> 53.9 ms, profiles/r03/sparse_knobs.log) */
constexpr int kChunkParticlesMin = 1024;
/* empty lanes that trigger a refill.  With the facet loop as lean as it is now a
 * wave does best refilling (almost) as a whole: its particles come from one tile and
 * start together, so their loads and LDS atomics stay close, which is worth more than
 * the idle lanes of the last facets (56: stream deck -6 %, csp -2 % against 32;
 * 56...64 are level) */
constexpr int kStreamRefillMin = 56;
/* facet crossings per STREAM pass: 64 (-1..2 % against 16), and 128 where the step's histories
 * end their flight under the window they start in (the step before needed one stream pass: csp
 * 400^2 with its ~63 facets per flight) -- the lanes of a wave that are through then wait for
 * its longest flight once, not for two passes' longest; where histories change windows anyway
 * (the stream deck: nine passes) 128 costs 4 % and 32 costs 3 (profiles/r04/experiments/
 * stream_knobs.log).  What a history computes does not depend on it. */
constexpr int kStreamRepeat = 64;
constexpr int kStreamRepeatOnePass = 128;
/* -DNEUTRAL_NO_CARRIED_START: the A/B build in which every instantiation looks up and draws in
 * the stream kernel's refill (tiled_uses_carried) */
#ifdef NEUTRAL_NO_CARRIED_START
constexpr bool kCarriedStart = false;
#else
constexpr bool kCarriedStart = true;
#endif
constexpr int kSortBlock = 256;
/* counting sort: records one workgroup histograms and places at a time, and the
 * largest number of buckets (tiles + 1) it keeps in LDS (count + base: 64 KB);
 * meshes with more tiles than that place with one global atomic per record */
constexpr int kSortItems = 16;
constexpr int kSortSegment = kSortBlock * kSortItems;
constexpr int kSortLdsBins = 8192;

enum Ctrl : int {
  kCtrlChunkHead = 0,
  kCtrlNumChunks = 1,
  kCtrlCollideCount = 2,
  kCtrlActive = 3,
  kCtrlMigrants = 4,   /* histories the last stream pass handed to the next one */
  kCtrlPassesUsed = 5, /* stream passes of this step that had work */
  kCtrlWindowed = 6,   /* chunks of this pass that stream under an LDS window */
  kCtrlEmigrants = 7,  /* decomposed mesh: histories waiting to be sent to another rank */
  kCtrlFirstInactive = 8, /* first slot of the particles that were dead when the step began
                             (pass 0 carries them over behind the live ones) */
  kCtrlLive = 9,       /* histories of this launch of the stream kernel that have not ended in
                          it yet (its workgroups leave when it is zero) */
  kCtrlHops = 10,      /* histories handed to another tile's queue inside the stream kernel */
  kCtrlOverflows = 11, /* ... that found the queue full and wait for the next pass instead */
  kCtrlBatches = 12,   /* claims of a workgroup on a tile's queue */
  kCtrlIdlePolls = 13, /* sweeps of a workgroup that found nothing to take */
};
/* ---- the asynchronous tile queue (TiledArgs::queue_entries) ---- */
constexpr unsigned kQueueEmpty = 0xFFFFFFFFu; /* a place that is reserved but not written yet */
/* Which queue a workgroup takes from, in this order:
 *   - the queue of the tile its window is centred on, while kQueueStayMin histories wait there
 *     (no window to flush and move);
 *   - a fresh chunk;
 *   - the first tile -- counted on from its own, every workgroup a different way round -- where
 *     a workgroup's worth waits (kQueueBatchMin: one history per lane), else the fullest.
 * Two policies that were tried first (profiles/r04/experiments/queue_policy_ab.log): every
 * workgroup to the FULLEST queue -- all 256 sweep at about the same time, agree on the tile and
 * split it into scraps: stream 4000^2 at 75 ms against the pass mechanism's 36, 57 % of the
 * workgroup time spent polling; and the queue whose head is OLDEST (entries tagged with how much
 * of their step lies behind them), which herds worse: 242 ms. */
constexpr unsigned kQueueStayMin = 256;
constexpr unsigned kQueueBatchMin = 1024;
constexpr unsigned kQueueBatchMax = 4096; /* histories a workgroup claims at a time: four per
                                             lane, so that lanes whose history was short are
                                             refilled instead of waiting for their wave's longest */
/* polls of a workgroup that finds nothing to do while histories are still in flight elsewhere
 * (each a sweep over the tiles' counters and a sleep: several microseconds) before it gives up
 * and reports the launch as aborted -- seconds; a launch lasts milliseconds */
constexpr unsigned kQueueIdlePollsMax = 1u << 20;
/* A history of a chunk WITHOUT a window (a sparse tile, the small tail of a dense one)
 * runs on global atomics.  When other chunks of the pass do have windows it is handed
 * to the next pass after this many facet crossings (about one window's worth), like a
 * history that left its window: the next sort may put it into a tile that is worth a
 * window, and no pass has to wait for a lone history that crosses thousands of cells
 * (the reference's stream deck as shipped: 7 000 facets per history and step). */
constexpr unsigned kUnwindowedBudget = 160;
/* a history leaves its window for another pass only if about this many facet
 * crossings still lie ahead; shorter tails finish with global atomics */
constexpr double kMigrateMinFacets = 8.0;
constexpr int kMaxStreamPasses = 256;

/* ---- 1. counting sort of the live records by tile ----------------------------------- */

constexpr unsigned kNoBucket = 0xFFFFFFFFu;

/* bucket of a record in this pass, from its 4-byte summary: its tile when it takes
 * part (pass 0: every live particle; later passes: the migrants of the pass before);
 * in pass 0 the dead go to the last bucket (they are carried over behind the live
 * ones); in later passes everything else stays where it is */
__device__ __forceinline__ unsigned sort_bucket(const TiledArgs& t, unsigned summary) {
  const int state = summary_state(summary);
  /* (tile, reach class): the classes of a tile follow each other, shortest flights first) */
  /* (longest flights first: the chunks of a tile are handed out in this order, and the
   * long ones are the ones worth starting early -- stream 4000^2 / 1e6 49.5 -> 47.9 ms,
   * profiles/r03/experiments/reach_class_ab.log) */
  const unsigned cls = 3u - summary_reach(summary);
  const unsigned live = (t.reach_classes > 1)
                            ? summary_tile(summary) * (unsigned)t.reach_classes + cls
                            : summary_tile(summary);
  if (t.pass == 0) {
    /* (a slot whose particle was sent to another rank is not carried over at all) */
    return (state == kRecGone) ? kNoBucket : (state != kRecDead) ? live : (unsigned)t.nsort;
  }
  return (state == kRecMigrate) ? live : kNoBucket;
}

/* the summaries a pass sorts: last step's in pass 0, this step's (written by the
 * stream kernel next to the records it hands on) afterwards */
__device__ __forceinline__ const unsigned* sort_summaries(const TiledArgs& t) {
  return (t.pass == 0) ? t.info_in : t.info_out;
}

/* a later pass with nobody to move: its kernels return at once */
__device__ __forceinline__ bool pass_is_empty(const TiledArgs& t) {
  return t.pass > 0 && t.ctrl[kCtrlMigrants] == 0;
}
/* an earlier launch of this step dropped histories (a defect the host reports: their summaries
 * were never written): nothing more of the step runs on what it left behind */
__device__ __forceinline__ bool step_is_broken(const SolveArgs& a) {
  return a.counters->aborted != 0u;
}

/* histogram: tile_count[b] += records of bucket b (tile_count is zero on entry:
 * tile_scan_kernel clears what it has consumed) */
__global__ __launch_bounds__(kSortBlock) void tile_count_kernel(SolveArgs a, TiledArgs t) {
  extern __shared__ unsigned s_bins[];
  if (pass_is_empty(t) || (t.pass > 0 && step_is_broken(a))) {
    return;
  }
  const int nbins = t.nsort + 1;
  const bool in_lds = nbins <= kSortLdsBins;
  if (in_lds) {
    for (int b = threadIdx.x; b < nbins; b += kSortBlock) {
      s_bins[b] = 0;
    }
    __syncthreads();
  }
  const unsigned* info = sort_summaries(t);
  const long long base = (long long)blockIdx.x * kSortSegment;
#pragma unroll 4
  for (int k = 0; k < kSortItems; ++k) {
    const long long i = base + (long long)k * kSortBlock + threadIdx.x;
    const unsigned b = (i < t.sort_end) ? sort_bucket(t, info[i]) : kNoBucket;
    if (in_lds) {
      if (b != kNoBucket) {
        atomicAdd(&s_bins[b], 1u);
      }
    } else {
      /* (more buckets than fit the LDS: one global atomic per record -- except for the dead,
       * who all share ONE bucket: a wave counts its own and adds once, or a step in which
       * most particles are dead queues millions of adds on a single word) */
      const bool is_dead = (b == (unsigned)t.nsort);
      const unsigned long long m_dead = __ballot(is_dead);
      if (b != kNoBucket && !is_dead) {
        atomicAdd(&t.tile_count[b], 1u);
      }
      if (m_dead != 0 && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m_dead) - 1u) {
        atomicAdd(&t.tile_count[t.nsort], (unsigned)__popcll(m_dead));
      }
    }
  }
  if (in_lds) {
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += kSortBlock) {
      const unsigned c = s_bins[b];
      if (c) {
        atomicAdd(&t.tile_count[b], c);
      }
    }
  }
}

/* single workgroup: tile_offset[k] = first sorted position of bucket k, for
 * k = 0..nsort+1 (empty tiles included); tile_cursor = the same, consumed by the
 * placement; tile_count is cleared for the next pass */
__global__ __launch_bounds__(1024) void tile_scan_kernel(TiledArgs t) {
  __shared__ unsigned s_part[1024];
  const int tid = threadIdx.x;
  const int n = t.nsort + 2;
  const int per = (n + 1023) / 1024;
  const int lo = (tid * per < n) ? tid * per : n;
  const int hi = (lo + per < n) ? lo + per : n;
  unsigned sum = 0;
  for (int i = lo; i < hi; ++i) {
    sum += (i <= t.nsort) ? t.tile_count[i] : 0u;
  }
  s_part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) { /* Hillis-Steele inclusive scan */
    const unsigned c0 = (tid >= off) ? s_part[tid - off] : 0;
    __syncthreads();
    s_part[tid] += c0;
    __syncthreads();
  }
  if (tid == 0) {
    t.ctrl[kCtrlWindowed] = 0; /* tile_chunks_kernel counts this pass's windowed chunks */
  }
  unsigned run = s_part[tid] - sum; /* exclusive */
  for (int i = lo; i < hi; ++i) {
    const unsigned c = (i <= t.nsort) ? t.tile_count[i] : 0u;
    t.tile_offset[i] = run;
    t.tile_cursor[i] = run;
    if (i <= t.nsort) {
      t.tile_count[i] = 0;
    }
    run += c;
  }
}

/* placement: order[position] = record index, positions of a bucket contiguous.  A
 * workgroup histograms its segment in LDS, reserves its part of every bucket it
 * holds with ONE returning atomic per bucket, and ranks its records inside the
 * reservation with LDS atomics.  (The order inside a tile is whatever the atomics
 * make it: nothing depends on it -- histories are independent and the tally is a
 * sum.) */
/* Pass 0 also makes every live history's first draw of the timestep (TiledArgs::carried_in):
 * Threefry and the logarithm, 250 vector instructions per record, in a kernel that otherwise
 * waits on atomics -- instead of at the head of the stream kernel's refill chain. */
__device__ __forceinline__ void draw_for_slot(const SolveArgs& a, const TiledArgs& t, long long slot) {
  t.carried_in[slot].minus_log_rn0 = draw_first_flight(a.pid_base + (uint64_t)t.id_in[slot], a.master_key);
}

__global__ __launch_bounds__(kSortBlock) void tile_scatter_kernel(SolveArgs a, TiledArgs t) {
  extern __shared__ unsigned s_bins[]; /* count/rank [nbins], base [nbins] */
  if (pass_is_empty(t) || (t.pass > 0 && step_is_broken(a))) {
    return;
  }
  const bool draws = t.pass == 0 && t.carried != 0;
  const int nbins = t.nsort + 1;
  const bool in_lds = nbins <= kSortLdsBins;
  unsigned* s_rank = s_bins;
  unsigned* s_base = s_bins + nbins;
  const unsigned* info = sort_summaries(t);
  const long long base = (long long)blockIdx.x * kSortSegment;
  unsigned bucket[kSortItems];
#pragma unroll
  for (int k = 0; k < kSortItems; ++k) {
    const long long i = base + (long long)k * kSortBlock + threadIdx.x;
    bucket[k] = (i < t.sort_end) ? sort_bucket(t, info[i]) : kNoBucket;
  }
  if (!in_lds) {
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
      /* (the dead share one bucket: a wave reserves its records' places with one atomic) */
      const bool is_dead = (bucket[k] == (unsigned)t.nsort);
      const unsigned long long m_dead = __ballot(is_dead);
      unsigned dead_base = 0;
      if (m_dead != 0) {
        const unsigned leader = (unsigned)__ffsll((long long)m_dead) - 1u;
        if ((threadIdx.x & 63) == leader) {
          dead_base = atomicAdd(&t.tile_cursor[t.nsort], (unsigned)__popcll(m_dead));
        }
        dead_base = __shfl(dead_base, (int)leader, 64);
      }
      if (bucket[k] != kNoBucket) {
        const unsigned pos = is_dead ? dead_base + (unsigned)lane_rank(m_dead)
                                     : atomicAdd(&t.tile_cursor[bucket[k]], 1u);
        t.order[pos] = (unsigned)(base + (long long)k * kSortBlock + threadIdx.x);
        if (draws && !is_dead) {
          draw_for_slot(a, t, base + (long long)k * kSortBlock + threadIdx.x);
        }
      }
    }
    return;
  }
  for (int b = threadIdx.x; b < nbins; b += kSortBlock) {
    s_rank[b] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kSortItems; ++k) {
    if (bucket[k] != kNoBucket) {
      atomicAdd(&s_rank[bucket[k]], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += kSortBlock) {
    const unsigned c = s_rank[b];
    if (c) {
      s_base[b] = atomicAdd(&t.tile_cursor[b], c);
      s_rank[b] = 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kSortItems; ++k) {
    if (bucket[k] != kNoBucket) {
      const unsigned pos = s_base[bucket[k]] + atomicAdd(&s_rank[bucket[k]], 1u);
      t.order[pos] = (unsigned)(base + (long long)k * kSortBlock + threadIdx.x);
    }
  }
  if (draws) {
    /* (a loop, not sixteen unrolled copies of Threefry: the kernel's code stays small) */
#pragma unroll 1
    for (int k = 0; k < kSortItems; ++k) {
      if (bucket[k] != kNoBucket && bucket[k] != (unsigned)t.nsort) {
        draw_for_slot(a, t, base + (long long)k * kSortBlock + threadIdx.x);
      }
    }
  }
}

/* pass 0: the records that sit the step out (the dead the sort found) are carried over
 * behind the live ones, which the stream kernel writes in tile order; and the ones carried
 * over a step ago are copied across to this buffer, where they stay (TiledArgs: graveyard) */
__global__ __launch_bounds__(kSortBlock) void copy_inactive_kernel(SolveArgs a, TiledArgs t) {
  if (a.abort_flag && *a.abort_flag) {
    return; /* (an abandoned attempt leaves slot_of_id as the last finished step made it) */
  }
  const unsigned first_inactive = t.tile_offset[t.nsort];
  const unsigned ncarried = (unsigned)t.sort_end - first_inactive;
  /* (a grid of a few thousand workgroups strides over what there is to carry -- known on the
   * device only -- instead of one workgroup per 256 particles of the store returning at once:
   * 390 625 dispatches at 1e8) */
  const unsigned total = ncarried + (unsigned)(t.mirror_end - t.sort_end);
  for (unsigned u = blockIdx.x * kSortBlock + threadIdx.x; u < total; u += gridDim.x * kSortBlock) {
  if (u < ncarried) {
    const unsigned j = first_inactive + u;
    const unsigned src = t.order[j];
    const ParticleRec r = t.rec_in[src];
    t.rec_out[j] = r;
    t.info_out[j] = t.info_in[src];
    if (a.decomposed) {
      t.id_out[j] = t.id_in[src];
    } else {
      /* (in lazy mode too: this is the last time the record moves, and a later eager step
       * writes slot_of_id only for what it places itself) */
      t.slot_of_id[r.id] = j;
    }
  } else {
    const unsigned m = (unsigned)t.sort_end + (u - ncarried);
    if (m < (unsigned)t.mirror_end) {
      t.rec_out[m] = t.rec_in[m]; /* same slot: slot_of_id stays */
      t.info_out[m] = t.info_in[m];
    }
  }
  }
}

/* after the last pass: the ids (positions in rec) of the histories suspended at
 * a collision, for the collision kernel */
__global__ __launch_bounds__(kSortBlock) void collect_suspended_kernel(SolveArgs a, TiledArgs t,
                                                                       const unsigned* info) {
  /* one queue reservation per segment of kSortSegment records: a returning atomic on a
   * single word takes ~100 ops/us, and every history of a collision-only deck is
   * suspended (scatter 1e8: one per 256 records cost 3.9 ms of atomics) */
  __shared__ unsigned s_count[kSortItems][kSortBlock / 64];
  __shared__ unsigned s_base;
  if (step_is_broken(a)) {
    return;
  }
  const long long base = (long long)blockIdx.x * kSortSegment;
  const int wave = threadIdx.x >> 6;
  unsigned mine = 0; /* bit k: record base + k * kSortBlock + threadIdx.x is suspended */
#pragma unroll
  for (int k = 0; k < kSortItems; ++k) {
    const long long i = base + (long long)k * kSortBlock + threadIdx.x;
    const bool susp = i < t.sort_end && summary_state(info[i]) == kRecCollide;
    const unsigned long long m = __ballot(susp);
    mine |= susp ? (1u << k) : 0u;
    if ((threadIdx.x & 63) == 0) {
      s_count[k][wave] = (unsigned)__popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0; /* exclusive prefix over (item, wave), in place */
    for (int k = 0; k < kSortItems; ++k) {
      for (int w = 0; w < kSortBlock / 64; ++w) {
        const unsigned c = s_count[k][w];
        s_count[k][w] = total;
        total += c;
      }
    }
    s_base = total ? atomicAdd(&t.ctrl[kCtrlCollideCount], total) : 0u;
  }
  __syncthreads();
  if (mine) {
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
      const bool susp = ((mine >> k) & 1u) != 0;
      const unsigned long long m = __ballot(susp);
      if (susp) {
        const unsigned slot = (unsigned)(base + (long long)k * kSortBlock + threadIdx.x);
        t.collide_queue[s_base + s_count[k][wave] + (unsigned)lane_rank(m)] = slot;
        if (t.mark_suspended) {
          /* (whose history the collision stage ends, and exports: the pass over the ids that
           * runs beside it leaves these alone.  Marked here -- a random 4-byte read of the id
           * per suspended record -- or by the stream kernel where it suspends: level, same box) */
          const unsigned id = t.rec_out[slot].id;
          atomicOr(&t.susp_ids[id >> 5], 1u << (id & 31u));
        }
      }
    }
  }
}

/* SoA store -> records, in id order */
__global__ __launch_bounds__(kSortBlock) void import_records_kernel(ParticleView p, ParticleRec* rec,
                                                                    unsigned* info,
                                                                    unsigned* slot_of_id,
                                                                    unsigned* ids, int tiles_x,
                                                                    int tile_shift, int x_off,
                                                                    int y_off, int n) {
  const int i = blockIdx.x * kSortBlock + threadIdx.x;
  if (i < n) {
    ParticleRec r;
    r.x = p.x[i];
    r.y = p.y[i];
    r.omega_x = p.omega_x[i];
    r.omega_y = p.omega_y[i];
    r.energy = p.energy[i];
    r.weight = p.weight[i];
    r.dt_to_census = p.dt_to_census[i];
    r.mfp_to_collision = p.mfp_to_collision[i];
    r.cellx = p.cellx[i];
    r.celly = p.celly[i];
    r.id = (unsigned)i;
    r.dead = p.dead[i];
    rec[i] = r;
    slot_of_id[i] = (unsigned)i;
    ids[i] = (unsigned)i;
    info[i] = slot_summary(r.dead ? kRecDead : kRecIdle, r.cellx - x_off, r.celly - y_off, tiles_x,
                           tile_shift);
  }
}

/* carried_in[slot].micro for every live record of rec_in (TiledArgs::carried_in): what the stream kernel's
 * carried start takes for granted, made good after an import or a rebuilt table view.  Plain
 * bisection -- the same unique bracket as the indexed search, hence the same value -- so that
 * nothing here depends on the cached indexes. */
__global__ __launch_bounds__(kSortBlock) void refresh_micro_kernel(SolveArgs a, TiledArgs t) {
  const long long i = (long long)blockIdx.x * kSortBlock + threadIdx.x;
  if (i >= t.sort_end) {
    return;
  }
  const int state = summary_state(t.info_in[i]);
  if (state == kRecDead || state == kRecGone) {
    return;
  }
  const double energy = t.rec_in[i].energy;
  const int ind = cs_bracket(a.scatter_keys, a.scatter_n, energy);
  t.carried_in[i].micro = a.checked ? cs_interpolate<true>(a.scatter_keys, a.scatter_values, ind, energy)
                            : cs_interpolate<false>(a.scatter_keys, a.scatter_values, ind, energy);
}

/* records -> SoA store.  A direct scatter (each record to the eleven arrays at its
 * id) writes 8 or 4 bytes into eleven different 64-B sectors per particle; going by
 * id instead costs one scattered 4-B write per particle (slot_of_id, by the kernel that
 * places the record: pass 0 of the stream kernel, copy_inactive, the import), one
 * random 80-B record read, and eleven fully coalesced stores. */
__global__ __launch_bounds__(kSortBlock) void export_records_kernel(const ParticleRec* rec,
                                                                    const unsigned* slot_of_id,
                                                                    ParticleView p, int n,
                                                                    const int* abort_flag,
                                                                    const unsigned* first_inactive,
                                                                    unsigned final_from,
                                                                    const unsigned* skip_ids,
                                                                    int partial_ok) {
  if (abort_flag && *abort_flag) {
    return; /* the step's kernels have done nothing: rec holds an older step */
  }
  /* (a grid smaller than the ids strides over them: the pass beside the collision stage) */
  for (long long kk = (long long)blockIdx.x * kSortBlock + threadIdx.x; kk < n; kk += (long long)gridDim.x * kSortBlock) {
    const int k = (int)kk;
    if (skip_ids && ((skip_ids[k >> 5] >> (k & 31)) & 1u)) {
      continue; /* in the collision stage's hands (it may be rewriting the record right now) */
    }
    const unsigned slot = slot_of_id[k];
    /* (the boundary as the step's sort found it, on the device -- or, written back on demand,
     * as the host remembers it from when the arrays were last current) */
    if (slot >= (first_inactive ? *first_inactive : final_from)) {
      continue; /* dead since before the step began: the arrays have its final state, and the
                 random access to its record -- what this pass is bound by -- is saved */
    }
    /* What every history changes sits in the record's first 48 bytes (ParticleRec): position, the
     * two clocks, cell, state.  Direction, energy, weight and the death flag change only in a
     * collision or a reflection, and the state word says whether one happened (kRecChanged): a
     * record without it -- nine in ten in csp -- is written back from its first three quads (one
     * 64-byte sector for every other record instead of two) into six of the eleven arrays; the other
     * five hold what they should already.  partial_ok: the arrays were current as the step
     * began (else every field is written). */
    const rec_quad* const q = (const rec_quad*)__builtin_assume_aligned(&rec[slot], 16);
    ParticleRec r;
    rec_quad* const rq = (rec_quad*)&r;
    rq[0] = q[0];
    rq[1] = q[1];
    rq[2] = q[2];
    /* (streamed out once: non-temporal stores keep them from evicting the half-read lines
     * of the records, -15 % on this pass: tools/micro/export_probe.hip) */
    __builtin_nontemporal_store(r.x, &p.x[k]);
    __builtin_nontemporal_store(r.y, &p.y[k]);
    __builtin_nontemporal_store(r.dt_to_census, &p.dt_to_census[k]);
    __builtin_nontemporal_store(r.mfp_to_collision, &p.mfp_to_collision[k]);
    __builtin_nontemporal_store(r.cellx, &p.cellx[k]);
    __builtin_nontemporal_store(r.celly, &p.celly[k]);
    if (!partial_ok || (r.dead & kRecChanged)) {
      rq[3] = q[3];
      rq[4] = q[4];
      __builtin_nontemporal_store(r.omega_x, &p.omega_x[k]);
      __builtin_nontemporal_store(r.omega_y, &p.omega_y[k]);
      __builtin_nontemporal_store(r.energy, &p.energy[k]);
      __builtin_nontemporal_store(r.weight, &p.weight[k]);
      __builtin_nontemporal_store((record_state(r.dead) == kRecDead) ? 1 : 0, &p.dead[k]);
    }
  }
}

/* single workgroup: the chunk list, from the tile populations.
 *
 * A tile with at least window_min_particles particles is cut into chunks of its own
 * (they stream with the LDS window centred on the tile).  Tiles below that are not
 * worth a window, so they need no chunk of their own either: consecutive sparse
 * tiles -- contiguous in the sorted order -- are MERGED into un-windowed chunks of
 * up to chunk_particles.  Without that a sparse problem (the reference's shipped
 * decks: 4000^2 cells, 1e6 particles, 16 per tile) hands 1024-thread workgroups
 * sixteen particles at a time (csp default deck: 822 -> see DESIGN.md).
 * Every thread owns a contiguous range of tiles sized to hold about one chunk of
 * particles; a run of sparse tiles ends at the range's end. */
__global__ __launch_bounds__(1024) void tile_chunks_kernel(TiledArgs t) {
  __shared__ unsigned s_chunks[1024];
  /* the bucket offsets, staged once (coalesced) when they fit: every thread then sweeps
   * its range of tiles twice, one dependent read per tile -- from global memory that was
   * 33 us per launch on the shipped csp deck, 121 launches per ten steps */
  constexpr int kStagedOffsets = 8192;
  __shared__ unsigned s_offset[kStagedOffsets + 2];
  const int tid = threadIdx.x;
  const bool staged = t.nsort + 1 <= kStagedOffsets;
  if (staged) {
    for (int i = tid; i <= t.nsort + 1; i += 1024) {
      s_offset[i] = t.tile_offset[i];
    }
    __syncthreads();
  }
  const unsigned* offset = staged ? s_offset : t.tile_offset;
  const unsigned nactive = offset[t.nsort];
  const int classes = t.reach_classes; /* (a tile's buckets follow each other) */
  long long per = (t.ntiles + 1023) / 1024;
  if (nactive > 0) {
    const long long per_sparse =
        ((long long)t.chunk_particles * (long long)t.ntiles + nactive - 1) / (long long)nactive;
    per = (per_sparse > per) ? per_sparse : per;
    /* (... but no thread sweeps more than kSweepTilesMax tiles: with a few hundred
     * histories left in a late pass of a sparse deck, one thread walked the whole mesh,
     * twice, while 1023 waited: 250 us per launch on the shipped csp deck, as long as the
     * stream kernel it feeds.  The chunks of such a pass are smaller than a chunk, which
     * costs nothing: they are a wave's worth of work each, on workgroups of their own) */
    constexpr long long kSweepTilesMax = 32;
    const long long floor_per = (t.ntiles + 1023) / 1024;
    const long long cap = (floor_per > kSweepTilesMax) ? floor_per : kSweepTilesMax;
    per = (per > cap) ? cap : per;
  }
  per = (per > t.ntiles) ? t.ntiles : per;
  const long long lo_ll = (long long)tid * per;
  const int lo = (lo_ll < t.ntiles) ? (int)lo_ll : t.ntiles;
  const int hi = (lo_ll + per < t.ntiles) ? (int)(lo_ll + per) : t.ntiles;
  const unsigned cp = (unsigned)t.chunk_particles;

  /* sweep 1 counts this range's chunks, sweep 2 (after the scan) writes them */
  unsigned nch = 0;
  unsigned chunk = 0;
  unsigned nwindowed = 0;
  for (int sweep = 0; sweep < 2; ++sweep) {
    unsigned run_begin = 0, run_end = 0; /* pending run of sparse tiles: [begin, end) */
    auto emit = [&](unsigned begin, unsigned end, unsigned tile, unsigned windowed) {
      if (end <= begin) {
        return;
      }
      /* equal parts (a tile of chunk_particles + 200 becomes two halves, not a full
       * chunk and a tail too small for a window) */
      const unsigned parts = (end - begin + cp - 1) / cp;
      const unsigned size = (end - begin + parts - 1) / parts;
      for (unsigned b = begin; b < end; b += size) {
        const unsigned e = (b + size < end) ? b + size : end;
        if (sweep == 0) {
          nch++;
        } else {
          const unsigned w = (windowed && (e - b) >= (unsigned)t.window_min_particles) ? 1u : 0u;
          if (chunk < (unsigned)t.max_chunks) {
            t.chunks[chunk] = make_uint4(b, e, tile, w);
          }
          nwindowed += w;
          chunk++;
        }
      }
    };
    for (int i = lo; i < hi; ++i) {
      const unsigned begin = offset[i * classes];
      const unsigned end = offset[(i + 1) * classes];
      if (end - begin >= (unsigned)t.window_min_particles) {
        emit(run_begin, run_end, 0u, 0u);
        run_begin = run_end = end;
        emit(begin, end, (unsigned)i, 1u);
      } else {
        if (run_end == run_begin) {
          run_begin = begin;
        }
        run_end = end;
      }
    }
    emit(run_begin, run_end, 0u, 0u);
    if (sweep == 0) {
      s_chunks[tid] = nch;
      __syncthreads();
      for (int off = 1; off < 1024; off <<= 1) { /* Hillis-Steele inclusive scan */
        const unsigned c0 = (tid >= off) ? s_chunks[tid - off] : 0;
        __syncthreads();
        s_chunks[tid] += c0;
        __syncthreads();
      }
      chunk = s_chunks[tid] - nch; /* exclusive */
    }
  }
  if (nwindowed) {
    atomicAdd(&t.ctrl[kCtrlWindowed], nwindowed); /* (zeroed by tile_scan_kernel) */
  }
  if (tid == 1023) {
    t.ctrl[kCtrlNumChunks] = s_chunks[1023];
    t.ctrl[kCtrlActive] = nactive;
    t.ctrl[kCtrlLive] = nactive; /* (every one of them ends in this launch, or waits in a queue of it) */
    t.ctrl[kCtrlChunkHead] = 0;
    t.ctrl[kCtrlCollideCount] = 0;
    t.ctrl[kCtrlMigrants] = 0;
    if (t.pass == 0) {
      t.ctrl[kCtrlPassesUsed] = 1;
      t.ctrl[kCtrlEmigrants] = 0;
      t.ctrl[kCtrlFirstInactive] = offset[t.nsort];
    } else if (nactive > 0) {
      t.ctrl[kCtrlPassesUsed] = (unsigned)t.pass + 1u;
    }
  }
}

/* One workgroup per tile: does every cell of the tile's tally window (the W x W cells the
 * stream kernel centres on it) AND of the ring of cells around it hold the same density,
 * bit for bit?  Then a history that leaves a cell inside the window enters a cell of the
 * density it has, and its crossing skips the load and the compare (neutral_history.h:
 * WindowCellTallyT::uniform).  Asked every step: the mesh is the caller's. */
__global__ __launch_bounds__(kSortBlock) void tile_uniform_kernel(SolveArgs a, TiledArgs t, int window_cells) {
  __shared__ int s_differs;
  if (threadIdx.x == 0) {
    s_differs = 0;
  }
  __syncthreads();
  const int tile = (int)blockIdx.x;
  const int margin = (window_cells - (1 << t.tile_shift)) >> 1;
  int x0 = ((tile % t.tiles_x) << t.tile_shift) - margin - 1;
  int y0 = ((tile / t.tiles_x) << t.tile_shift) - margin - 1;
  int x1 = x0 + window_cells + 2; /* (exclusive) */
  int y1 = y0 + window_cells + 2;
  x0 = (x0 < 0) ? 0 : x0;
  y0 = (y0 < 0) ? 0 : y0;
  x1 = (x1 > a.nx) ? a.nx : x1;
  y1 = (y1 > a.ny) ? a.ny : y1;
  const int w = x1 - x0;
  const int n = w * (y1 - y0);
  const long long first = __double_as_longlong(a.density[(size_t)y0 * a.nx + x0]);
  bool differs = false;
  for (int i = threadIdx.x; i < n; i += kSortBlock) {
    const int y = y0 + i / w;
    const int x = x0 + i % w;
    differs |= (__double_as_longlong(a.density[(size_t)y * a.nx + x]) != first);
  }
  if (__ballot(differs) != 0 && (threadIdx.x & 63) == 0) {
    s_differs = 1; /* (benign race: every writer writes 1) */
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    t.tile_uniform[tile] = s_differs ? 0 : 1;
  }
}

/* Single workgroup: is every edge of both axes the host layer's formula, edge[pad + i] ==
 * edge_d * (double)(off + i), bit for bit?  (SolveArgs::edge_dx ...)  Asked every step: the
 * arrays are the caller's. */
__global__ __launch_bounds__(1024) void edges_check_kernel(SolveArgs a, TiledArgs t) {
  __shared__ int s_differs;
  if (threadIdx.x == 0) {
    s_differs = 0;
  }
  __syncthreads();
  bool differs = !(a.edge_dx > 0.0) || !(a.edge_dy > 0.0);
  for (int i = threadIdx.x; i <= a.nx; i += 1024) {
    differs |= __double_as_longlong(edge_from_formula(a.edge_dx, i + a.x_off)) !=
               __double_as_longlong(a.edgex[i + a.pad]);
  }
  for (int i = threadIdx.x; i <= a.ny; i += 1024) {
    differs |= __double_as_longlong(edge_from_formula(a.edge_dy, i + a.y_off)) !=
               __double_as_longlong(a.edgey[i + a.pad]);
  }
  if (__ballot(differs) != 0 && (threadIdx.x & 63) == 0) {
    s_differs = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *t.edges_computed = s_differs ? 0 : 1;
  }
}

/* ---- 2. streaming kernel with the LDS tally window ------------------------------ */

/* One workgroup per tile, ahead of every launch of the stream kernel: the places of the tile's
 * log that the last launch used are empty again, the log starts over.  (A launch that handed
 * nobody on -- most launches after the first of a step -- leaves nothing to clear.) */
__global__ __launch_bounds__(kSortBlock) void queue_reset_kernel(TiledArgs t) {
  const int tile = (int)blockIdx.x;
  unsigned used = t.queue_tail[tile];
  used = (used > t.queue_capacity) ? t.queue_capacity : used;
  unsigned* log = t.queue_entries + (size_t)tile * t.queue_capacity;
  for (unsigned i = threadIdx.x; i < used; i += kSortBlock) {
    log[i] = kQueueEmpty;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    t.queue_tail[tile] = 0u;
    t.queue_head[tile] = 0u;
  }
}

template <int kW>
__device__ __forceinline__ void flush_window(const SolveArgs& a, double* window, double* mesh,
                                             int ox, int oy) {
  /* row-contiguous: one wave instruction adds 64 consecutive cells (512 B) */
  constexpr int kS = kW + kWindowRowPad; /* (cells per row in LDS: neutral_history.h) */
  for (int i = threadIdx.x; i < kW * kW; i += kStreamBlock) {
    const int at = (i / kW) * kS + (i % kW);
    const double v = window[at];
    window[at] = 0.0;
    if (v != 0.0) {
      const int gx = ox + (i % kW);
      const int gy = oy + (i / kW);
      if (gx >= 0 && gx < a.nx && gy >= 0 && gy < a.ny) {
        unsafeAtomicAdd(&mesh[gy * a.nx + gx], v);
      }
    }
  }
}

/* bytes of the stream kernel's dynamic LDS in front of its control words: the window(s),
 * the staged cs index(es), rounded up to 16 */
/* (kWithIndex = false: an instantiation that starts histories from carried values looks
 * nothing up and stages no index) */
template <bool kSameTables, bool kWithIndex = true>
__host__ __device__ __forceinline__ size_t stream_lds_payload_bytes(const SolveArgs& a, int window_cells) {
  size_t lds = sizeof(double) * (size_t)window_cells;
  if (kWithIndex && a.scatter_index) {
    lds += sizeof(unsigned short) * (size_t)(a.scatter_index_n + 1);
  }
  if (kWithIndex && !kSameTables && a.absorb_index) {
    lds += sizeof(unsigned short) * (size_t)(a.absorb_index_n + 1);
  }
  return (lds + 15) & ~(size_t)15;
}
#ifdef NEUTRAL_PHASE_CLOCK
constexpr size_t kStreamLdsControlBytes = 64 + 16 * 9 * 8;
#else
constexpr size_t kStreamLdsControlBytes = 64;
#endif

/* more than kMigrateMinFacets facets ahead of it before the census, at the rate its
 * direction crosses cells? */
__device__ __forceinline__ bool far_to_go(const History& h, const TiledArgs& t) {
  const double ahead = h.speed * h.dt_to_census;
  const double facets_ahead = ahead * (fabs(h.omega_x) * t.cells_per_x +
                                       fabs(h.omega_y) * t.cells_per_y);
  return facets_ahead > kMigrateMinFacets;
}

/* -DNEUTRAL_PHASE_CLOCK: a build that says where the stream kernel's waves spend their time
 * (shader-clock ticks, summed over waves; tools/micro/phase_clock.py reads and prints them):
 * 0 taking work + barriers, 1 window flush and move, 2 refill (loads, prologue / resume),
 * 3 a stream pass outside its facet loop, 5 the facet loop, 6 census / end of a history (its
 * stores), 7 hand-offs (records of colliders and migrants).  A wave's stamps and sums live in
 * LDS behind the control words; whichever lane is the first active one where a stamp stands
 * makes it (the regions are divergent). */
#ifdef NEUTRAL_PHASE_CLOCK
__device__ unsigned long long g_phase_clock[8];
#define PHASE_DECL                                                                              \
  unsigned long long* const lds_ph =                                                            \
      (unsigned long long*)((char*)lds_ctl + 64) + (threadIdx.x >> 6) * 9;                      \
  if ((threadIdx.x & 63) == 0) {                                                                \
    lds_ph[0] = __builtin_readcyclecounter();                                                   \
    for (int k = 1; k <= 8; ++k) lds_ph[k] = 0;                                                 \
  }
#define PHASE(k)                                                                                \
  do {                                                                                          \
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {        \
      const unsigned long long ph_now = __builtin_readcyclecounter();                           \
      lds_ph[1 + (k)] += ph_now - lds_ph[0];                                                    \
      lds_ph[0] = ph_now;                                                                       \
    }                                                                                           \
  } while (0)
#define PHASE_REPORT                                                                            \
  if ((threadIdx.x & 63) == 0) {                                                                \
    for (int k = 0; k < 8; ++k) atomicAdd(&g_phase_clock[k], lds_ph[1 + k]);                    \
  }
extern "C" void neutral_hip_debug_phase_clock(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase_clock), sizeof(unsigned long long) * 8);
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clock), zero, sizeof(zero));
}
#else
#define PHASE_DECL
#define PHASE(k)
#define PHASE_REPORT
#endif

/* kQueues: the asynchronous tile queue is compiled in (TiledArgs::queue_entries: a property of
 * the kernel, like the flux and the decomposition -- merely carrying the queue code costs the
 * default instantiation 7 % of csp's stream stage in scalar and vector spills around the chunk
 * loop: profiles/r04/experiments/queue_policy_ab.log) */
template <bool kSameTables, bool kFlux, bool kDomain, bool kChecked, bool kQueues>
__global__ __launch_bounds__(kStreamBlock) void stream_kernel(SolveArgs a, TiledArgs t) {
  /* histories start from carried values (neutral_history.h: prologue_carried; the launcher sees
   * to it that they are valid: tiled_uses_carried): no lookup, no draw, no index in LDS */
  constexpr bool kCarried = kCarriedStart && kSameTables && !kDomain && !kQueues;
  constexpr int kW = WindowTallyT<kFlux, kCarried>::W; /* window edge; kWindows of them in LDS */
  constexpr int kWindows = kFlux ? 2 : 1;
  extern __shared__ double lds_raw[];
  constexpr int kWindowDoubles = kW * (kW + kWindowRowPad); /* a window in LDS: kW rows (neutral_history.h) */
  double* window = lds_raw;                                             /* kWindows of them */
  unsigned short* lds_index = (unsigned short*)(lds_raw + kWindows * kWindowDoubles);
  /* the workgroup's chunk bookkeeping lives BEHIND the window and the indexes, not in static
   * LDS in front of them: the window then starts at LDS address 0 and a cell's address is a
   * shift and a shift-add, without the add of the static variables' size (one vector
   * instruction per facet) */
  int* const lds_ctl = (int*)((char*)lds_raw + stream_lds_payload_bytes<kSameTables, !kCarried>(a, kWindows * kWindowDoubles));
  int& s_chunk = lds_ctl[0];
  int& s_end = lds_ctl[1];
  int& s_tile = lds_ctl[2];
  int& s_windowed = lds_ctl[3];
  int& s_cursor = lds_ctl[4];
  int& s_kind = lds_ctl[5];                      /* kWorkChunk / kWorkQueue / kWorkNone */
  unsigned& s_best = ((unsigned*)lds_ctl)[6];    /* the sweep over the tile queues: fullest ... */
  int& s_best_tile = lds_ctl[7];                 /* ... and which */

  if (a.abort_flag && *a.abort_flag) {
    return; /* the cached view of the cs tables is stale: the host re-runs the step */
  }
  clock_stamp_begin(a.counters);
  /* stage the cs index(es), zero the window */
  CsLookup<const unsigned short*> ix{nullptr, nullptr};
  {
    int used = 0;
    if (!kCarried && a.scatter_index) {
      for (int i = threadIdx.x; i <= a.scatter_index_n; i += kStreamBlock) {
        lds_index[i] = a.scatter_index[i];
      }
      ix.scatter_index = lds_index;
      used = a.scatter_index_n + 1;
    }
    if (!kCarried && !kSameTables && a.absorb_index) {
      for (int i = threadIdx.x; i <= a.absorb_index_n; i += kStreamBlock) {
        lds_index[used + i] = a.absorb_index[i];
      }
      ix.absorb_index = lds_index + used;
    }
    for (int i = threadIdx.x; i < kWindows * kWindowDoubles; i += kStreamBlock) {
      window[i] = 0.0;
    }
  }

  const int nchunks = (int)t.ctrl[kCtrlNumChunks];
  /* do SolveArgs::edge_dx / edge_dy reproduce the edge arrays this step?  (wave-uniform) */
  const bool edges_computed =
      t.edges_computed && __builtin_amdgcn_readfirstlane(*t.edges_computed) != 0;
  /* histories without a window move on after a window's worth of facets when the
   * pass has windows to offer (wave-uniform) */
  const bool budget_on = t.allow_migrate && (kQueues || t.ctrl[kCtrlWindowed] != 0);
  int cur_tile = -1; /* tile the LDS window is centred on (holds its partial sums) */
  int win_ox = 0;
  int win_oy = 0;
  WindowTallyT<kFlux, kCarried> tally{(lds_double*)window, 0, 0};

  unsigned nfacets = 0;     /* per lane */
  unsigned w_processed = 0; /* per wave (uniform) */
  unsigned w_census = 0;
  unsigned w_migrants = 0;
  unsigned w_emigrants = 0; /* histories that crossed into another rank's cells */

  History h;
  h.ev = kEvEnd;
  int pid = -1;

  /* the tile queues are in use (wave-uniform): migrants change tiles inside this launch */
  const bool queues = kQueues && t.allow_migrate;
  const unsigned qcap = t.queue_capacity;
  unsigned w_ended = 0;     /* histories this wave has ended for this launch since its last report */
  unsigned w_hops = 0;
  unsigned w_overflows = 0;
  unsigned wg_batches = 0;    /* (thread 0's: NeutralHipStepStats.stream_batches / stream_idle_polls) */
  unsigned wg_idle_polls = 0;
  enum : int { kWorkChunk = 0, kWorkQueue = 1, kWorkNone = 2 };

  PHASE_DECL
  for (;;) {
    if (queues && (threadIdx.x & 63) == 0 && w_ended) {
      atomicSub(&t.ctrl[kCtrlLive], w_ended);
    }
    w_ended = 0;
    /* ---- what next?  A queue with a workgroup's worth of waiting histories, else a fresh
     * chunk, else whatever waits in any queue; with nothing to take and histories still in
     * flight elsewhere (they may yet arrive here): look again; with none: done. ---- */
    unsigned polls = 0;
    for (;;) {
      __syncthreads(); /* the previous work is complete (also orders the staging above) */
      /* a claim on a tile's queue: places [head, head + m) of its log (thread 0) */
      auto claim = [&](int tile, unsigned at_least) -> bool {
        for (;;) {
          const unsigned head = __hip_atomic_load(&t.queue_head[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned tail = __hip_atomic_load(&t.queue_tail[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tail = (tail > qcap) ? qcap : tail;
          if (tail <= head || tail - head < at_least) {
            return false; /* (others were quicker) */
          }
          unsigned m = tail - head;
          m = (m > kQueueBatchMax) ? kQueueBatchMax : m;
          if (atomicCAS(&t.queue_head[tile], head, head + m) == head) {
            s_cursor = (int)head;
            s_end = (int)(head + m);
            s_tile = tile;
            s_windowed = 1;
            return true;
          }
        }
      };
      if (threadIdx.x == 0) {
        s_best = 0u;
        s_best_tile = -1;
        int kind = kWorkNone;
        /* stay where the window is, or take a fresh chunk: no sweep */
        if (queues && cur_tile >= 0 && claim(cur_tile, kQueueStayMin)) {
          kind = kWorkQueue;
          wg_batches++;
        }
        if (kind == kWorkNone &&
            __hip_atomic_load(&t.ctrl[kCtrlChunkHead], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nchunks) {
          const int c = (int)atomicAdd(&t.ctrl[kCtrlChunkHead], 1u);
          if (c < nchunks) {
            const uint4 ch = t.chunks[c];
            s_cursor = (int)ch.x;
            s_end = (int)ch.y;
            s_tile = (int)ch.z;
            s_windowed = (int)ch.w;
            kind = kWorkChunk;
          }
        }
        s_kind = kind;
      }
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(s_kind) != kWorkNone) {
        break;
      }
      if (!queues) {
        break; /* (no queues: out of chunks is out of work) */
      }
      /* sweep over the tiles' queues: the first tile after this workgroup's own where a
       * workgroup's worth waits; if there is none, the fullest */
      {
        const int start = (cur_tile >= 0) ? cur_tile : (int)((blockIdx.x * 2654435761u) % (unsigned)t.ntiles);
        unsigned mine = 0u;
        int mine_tile = -1;
        for (int tile = (int)threadIdx.x; tile < t.ntiles; tile += kStreamBlock) {
          unsigned tail = __hip_atomic_load(&t.queue_tail[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned head = __hip_atomic_load(&t.queue_head[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tail = (tail > qcap) ? qcap : tail;
          if (tail > head) {
            const unsigned waiting = tail - head;
            unsigned key;
            if (waiting >= kQueueBatchMin) {
              int ahead = tile - start;
              ahead += (ahead < 0) ? t.ntiles : 0;
              key = 0x80000000u | (unsigned)(t.ntiles - ahead);
            } else {
              const unsigned salt = ((unsigned)tile * 2654435761u + blockIdx.x * 40503u + polls * 9973u) >> 22;
              key = (waiting << 10) | salt;
            }
            if (key > mine) {
              mine = key;
              mine_tile = tile;
            }
          }
        }
        unsigned most = mine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned other = __shfl_xor(most, off, 64);
          most = (other > most) ? other : most;
        }
        const unsigned long long m_top = __ballot(most != 0u && mine == most);
        if (m_top != 0ull && __builtin_ctzll(m_top) == (int)(threadIdx.x & 63)) {
          atomicMax(&s_best, most);
        }
        __syncthreads();
        if (mine != 0u && mine == s_best) {
          s_best_tile = mine_tile; /* (equal keys: equally good, either will do) */
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        int kind = kWorkNone;
        if (s_best != 0u && claim(s_best_tile, 1u)) {
          kind = kWorkQueue;
        }
        s_kind = kind;
        if (kind == kWorkQueue) {
          wg_batches++;
        }
        /* nothing to take: is anything still in flight?  (s_chunk < 0: the workgroup leaves) */
        s_chunk = 0;
        if (kind == kWorkNone) {
          const unsigned live = __hip_atomic_load(&t.ctrl[kCtrlLive], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (live == 0u) {
            s_chunk = -1;
          } else if (s_best != 0u) {
            /* (a queue had something a moment ago: look again at once) */
          } else if (polls >= kQueueIdlePollsMax) {
            atomicAdd(&a.counters->aborted, 1u); /* (histories in flight that never arrive: a defect) */
            s_chunk = -1;
          } else {
            /* (idle workgroups back off: 255 of them sweeping every few microseconds is a
             * load on the L2 the working ones feel) */
            wg_idle_polls++;
            const unsigned nap = (polls < 5u) ? polls : 5u;
            for (unsigned k = 0; k < (1u << nap); ++k) {
              __builtin_amdgcn_s_sleep(64);
            }
          }
        }
      }
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(s_kind) != kWorkNone || __builtin_amdgcn_readfirstlane(s_chunk) < 0) {
        break;
      }
      polls++;
    }
    /* (values read from LDS arrive in vector registers: readfirstlane tells the
     * compiler they are wave-uniform, so the chunk bookkeeping and the window
     * origin stay on the scalar unit) */
    const int work = __builtin_amdgcn_readfirstlane(s_kind);
    PHASE(0);
    if (work == kWorkNone) {
      break;
    }
    const bool from_queue = (work == kWorkQueue);
    /* a fresh history (pass 0, from the sorted order): prologue, its place in rec_out is its
     * position; anything else resumes in place */
    const bool fresh = !from_queue && t.pass == 0;
    const int chunk_end = __builtin_amdgcn_readfirstlane(s_end);
    const int chunk_tile = __builtin_amdgcn_readfirstlane(s_tile);
    /* a dense tile's chunk (tile_chunks_kernel) */
    const bool windowed = (__builtin_amdgcn_readfirstlane(s_windowed) != 0);
    if (windowed && chunk_tile != cur_tile) {
      /* move the window: flush what the previous tile accumulated */
      if (cur_tile >= 0) {
        flush_window<kW>(a, window, a.tally, win_ox, win_oy);
        if (kFlux) {
          flush_window<kW>(a, window + kWindowDoubles, a.flux_tally, win_ox, win_oy);
        }
      }
      cur_tile = chunk_tile;
      /* the window reaches (W - T) / 2 cells beyond the T x T tile on every side */
      const int margin = (kW - (1 << t.tile_shift)) >> 1;
      win_ox = ((cur_tile % t.tiles_x) << t.tile_shift) - margin;
      win_oy = ((cur_tile / t.tiles_x) << t.tile_shift) - margin;
      __syncthreads();
    }
    PHASE(1);
    /* does every cell of the window, and around it, hold one density?  (wave-uniform) */
    const bool uniform_window =
        windowed && t.tile_uniform &&
        __builtin_amdgcn_readfirstlane((int)t.tile_uniform[cur_tile]) != 0;
    /* ... then every history of the chunk starts in a cell of that density (the chunk's records
     * were sorted by the tile of their cell: they are inside the tile, hence the window): one
     * scalar load per chunk instead of a load per history that waits for the record's */
    double window_density = 0.0;
    if (kCarried && uniform_window) {
      const int tile_x = (cur_tile % t.tiles_x) << t.tile_shift;
      const int tile_y = (cur_tile / t.tiles_x) << t.tile_shift;
      window_density = a.density[(size_t)tile_y * (size_t)a.nx + (size_t)tile_x];
    }
    /* may a history that leaves the window wait for another pass?  (all lanes or none: the
     * wave-uniform condition as a lane mask -- and a mask like any other to the compiler:
     * knowing it is all or nothing, it selects between the lanes' mask and zero with three
     * more scalar instructions per facet) */
    unsigned long long may_migrate =
        __builtin_amdgcn_readfirstlane((windowed && t.allow_migrate) ? 1 : 0) ? ~0ull : 0ull;
    asm volatile("" : "+s"(may_migrate));
    /* an un-windowed chunk sees a window that contains no cell */
    tally.ox = __builtin_amdgcn_readfirstlane(windowed ? win_ox : (1 << 30));
    tally.oy = __builtin_amdgcn_readfirstlane(windowed ? win_oy : (1 << 30));

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
      int park = kRecIdle; /* kRecCollide / kRecMigrate: this lane hands its history on */
      bool did_census = false;
      bool ended = false; /* this lane's history is over as far as this launch is concerned */
      if (n_empty >= t.refill_min || n_stream == 0) {
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
          /* wave-level bookkeeping (scalar registers, not one VGPR per counter) */
          if (fresh) {
            const int left = chunk_end - base;
            w_processed += (unsigned)((n_empty < left) ? n_empty : left);
          }
          if (!has && mine < chunk_end) {
            /* pass 0 reads last step's records through the sorted order and writes
             * this step's in tile order; later passes work on their migrants in place */
            unsigned src;
            if (from_queue) {
              /* place `mine` of the tile's log: reserved before it was claimed, written a
               * moment after it was reserved (bounded wait: a place never written is a defect,
               * and its history is reported, not waited for) */
              const unsigned* place = t.queue_entries + ((size_t)chunk_tile * qcap + (size_t)mine);
              unsigned spins = 0;
              do {
                src = __hip_atomic_load(place, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              } while (src == kQueueEmpty && ++spins < (1u << 22));
            } else {
              src = t.order[mine];
            }
            /* (a slot number that is none -- a place never written, or anything else that is not
             * a record of this store -- is reported and dropped, never dereferenced) */
            if (src >= (unsigned)a.nparticles) {
              atomicAdd(&a.counters->aborted, 1u);
              ended = true; /* (it leaves the launch's count of histories in flight: the other
                             * workgroups do not poll for it until they give up) */
            } else {
            pid = fresh ? mine : (int)src; /* this history's slot in rec_out */
            /* (asked for with the record, not after it: TiledArgs::carried_in) */
            double micro = 0.0, minus_log_rn0 = 0.0;
            if (kCarried) {
              const CarriedStart cs = fresh ? t.carried_in[src] : t.carried_out[src]; /* (one 16-byte load) */
              micro = cs.micro;
              minus_log_rn0 = cs.minus_log_rn0; /* (a migrant's: not looked at) */
            }
            bool bad_load = false;
            if (from_queue || (queues && !fresh)) {
              /* stored by another workgroup of this launch: around the L1 (neutral_history.h);
               * (a later pass's own migrants too: its lines' neighbours are changing hands) */
              load_record_through(h, a, t.rec_out[src], true);
              if (from_queue) {
                const int want_tile = ((h.celly - a.y_off) >> t.tile_shift) * t.tiles_x + ((h.cellx - a.x_off) >> t.tile_shift);
                /* (a record that does not belong to the queue it was taken from is a defect of
                 * the hand-off: reported and dropped, never streamed) */
                bad_load = (want_tile != chunk_tile) | (!kDomain && (h.id & ~kIdChanged) >= (unsigned)a.nparticles);
              }
            } else {
              load_record(h, a, fresh ? t.rec_in[src] : t.rec_out[src], !fresh);
            }
            if (fresh) {
              /* who lives in the slot / where the particle lives: what the write-back (or a
               * decomposed store's compaction) goes by, without reading 80-B records */
              t.id_out[pid] = h.id; /* (coalesced: next step's first draw goes by it) */
              if (!kDomain && t.slots_by_id) {
                t.slot_of_id[h.id] = (unsigned)pid;
              }
              if (kCarried) {
                t.carried_out[pid].micro = micro; /* (a streaming history keeps its energy) */
              }
            }
            if (bad_load) {
              atomicAdd(&a.counters->aborted, 1u);
              ended = true; /* (as above) */
            } else {
            if (kCarried) {
              if (fresh) {
                prologue_carried<kChecked>(h, a, micro, minus_log_rn0, uniform_window, window_density);
              } else {
                resume_carried<kChecked>(h, a, micro, uniform_window, window_density);
              }
            } else if (fresh) {
              prologue<kSameTables, kChecked>(h, a, ix);
            } else {
              resume<kSameTables, kChecked>(h, a, ix); /* a migrant: mid-history, no draw pending */
            }
            if (!kDomain) {
              /* a history in this kernel has drawn once (collisions happen elsewhere), and
               * saying so keeps the counter out of the facet loop's registers; only a
               * history that arrived from another rank can have collided already */
              h.counter = 1;
            }
            h.plain_div = 0;
            refresh_speed_reciprocal<kChecked>(h); /* no collision here: the speed stays */
            refresh_mfp_reciprocal<kChecked>(h);
            h.dep_rate = deposit_rate(h, a); /* ... and so do the weight and the heating factors */
            /* (wave-uniform branch: computed or loaded, the same bits) */
            if (edges_computed) {
              load_targets<true>(h, a);
            } else {
              load_targets<false>(h, a);
            }
            decide_carried(h);
            h.nevents++; /* (as decide() counts it) */
            has = true;
            if (h.ev == kEvCollision) {
              park = kRecCollide;
            }
            }
            }
          }
        }
        PHASE(2);
      } else if (has) {
        /* STREAM: up to kStreamRepeat facet crossings (the pass choice above costs
         * ~100 scalar instructions; histories cross ~60 facets in a row), or the
         * end of the history */
        if (h.ev == kEvFacet) {
          /* facets this lane crosses in this pass.  (It is the loop's trip count as well, and
           * stays the only counter: one kept apart from the compiler's own -- an opaque start
           * -- is one live register more than the loop has; scratch doubles and every trip
           * copies half the history, 126 -> 166 vector instructions) */
          int crossed = 0;
          /* the facet loop, compiled twice: for a window whose cells -- and the ring of cells
           * around it -- all hold one density (no density load, no compare while the history
           * is inside it: WindowCellTallyT<, true>), and for any other */
          h.m_x_facet = __builtin_amdgcn_ballot_w64(h.x_facet != 0); /* (decide_carried's) */
          /* (all lanes or none: the wave-uniform condition as a lane mask) */

          auto run_facets = [&](auto uniform_density, auto computed_edges) {
          constexpr bool kUniform = decltype(uniform_density)::value;
          constexpr bool kEdges = decltype(computed_edges)::value;
          /* One exit, at the bottom.  "Outside the window with a long way to go: continue in
           * the pass that centres a window on wherever the particle is by then" is asked
           * about the cell a crossing ENTERS, together with "is the next event another
           * facet" -- before the first trip for the cell the history starts in -- so a trip
           * has one place where lanes leave, and the window coordinates of the new cell are
           * worked out once: for that question and for the next trip's tally. */
          WindowCellTallyT<kFlux, kUniform, kCarried> cell_tally{
              tally.window, (unsigned)(h.cellx - a.x_off - tally.ox),
              (unsigned)(h.celly - a.y_off - tally.oy), 0ull};
          bool out_of_window = cell_tally.outside();
          cell_tally.m_outside = __builtin_amdgcn_ballot_w64(out_of_window);
          bool run = true;
          if ((cell_tally.m_outside & may_migrate) != 0) {
            asm volatile(""); /* (a branch the wave takes or skips, not a predicated block) */
            if (out_of_window) {
              run = !far_to_go(h, t);
            }
          }
          PHASE(3);
          if (run) {
#pragma unroll 1
            do {
              /* (tallies the cell it leaves: this one) */
              cross_facet<kChecked, true, kDomain ? 1 : 0, true, kEdges>(h, a, cell_tally);
              /* (counted in place, by hand: written as ++crossed the compiler compares the
               * old value, adds into a new register and copies it back) */
              asm("v_add_u32_e32 %0, 1, %0" : "+v"(crossed));
              if (kDomain) {
                /* the neighbour cell may belong to another rank: the history stops on the
                 * facet, before anything of that cell (edges, density) is looked at */
                if (outside_domain(h, a)) {
                  park = kRecEmigrate;
                  break;
                }
              }
              cell_tally.lx = (unsigned)(h.cellx - a.x_off - tally.ox);
              cell_tally.ly = (unsigned)(h.celly - a.y_off - tally.oy);
              out_of_window = cell_tally.outside();
              cell_tally.m_outside = __builtin_amdgcn_ballot_w64(out_of_window);
              /* (what ends the flight is named below the loop, once; `crossed` is the trip
               * count too: every lane in the loop started with it) */
              run = surely_next_is_facet(h) & (crossed < t.stream_repeat);
              /* (the scalar AND says whether its result is zero: one instruction and a branch) */
              if ((cell_tally.m_outside & may_migrate) != 0) {
                asm volatile("");
                if (out_of_window) {
                  run = run & !far_to_go(h, t);
                }
              }
            } while (run);
          }
          PHASE(5);
          };
          /* ... and for a mesh whose edges the device has found to follow the host layer's
           * formula this step (computed, not loaded) and for any other: four copies, one runs */
          if (uniform_window) {
            if (edges_computed) {
              run_facets(std::true_type{}, std::true_type{});
            } else {
              run_facets(std::true_type{}, std::false_type{});
            }
          } else {
            if (edges_computed) {
              run_facets(std::false_type{}, std::true_type{});
            } else {
              run_facets(std::false_type{}, std::false_type{});
            }
          }
          /* the event that ended the run of crossings (or another facet), from the
           * state the loop left: the comparisons next_is_facet() made, with names */
          decide_carried(h);
          park = (park == kRecIdle && h.ev == kEvCollision) ? (int)kRecCollide : park;
          /* a history that stopped in front of a facet, outside the window and with far to
           * go, moves on to the pass that centres a window on it: the one that left the loop
           * for that reason, and the one the loop's trip count stopped (it would leave on
           * the next pass's first trip) */
          if (windowed && t.allow_migrate && park == kRecIdle && h.ev == kEvFacet) {
            const unsigned lx = (unsigned)(h.cellx - a.x_off - tally.ox);
            const unsigned ly = (unsigned)(h.celly - a.y_off - tally.oy);
            if (!((lx < (unsigned)kW) & (ly < (unsigned)kW)) && far_to_go(h, t)) {
              park = kRecMigrate;
            }
          }
          /* facets are counted, and the event watchdog applied, once per pass (the
           * loop counter is scalar: per facet they cost three vector and half a
           * dozen scalar instructions) */
          nfacets += (unsigned)crossed;
          h.nevents += (unsigned)crossed;
          if (!windowed && budget_on && park == kRecIdle && h.ev == kEvFacet &&
              h.nevents >= kUnwindowedBudget) {
            /* (h.nevents counts from the load of this pass: prologue and resume zero it) */
            const double ahead = h.speed * h.dt_to_census;
            if (ahead * (fabs(h.omega_x) * t.cells_per_x + fabs(h.omega_y) * t.cells_per_y) >
                kMigrateMinFacets) {
              park = kRecMigrate;
            }
          }
          if (h.nevents > kMaxEventsPerHistory && h.ev == kEvFacet && park == kRecIdle) {
            atomicAdd(&a.counters->aborted, 1u);
            h.ev = kEvEnd; /* ended like a history whose time has run out, next pass */
          }
          PHASE(3);
        } else {
          if (h.ev == kEvCensus) {
            census_streamed<kChecked>(h, a, tally);
          }
          /* kEvEnd: the loop at omp3/neutral.c:134 exits */
          /* (a plain store: nobody picks this record up again in this launch.  Records that
           * change hands share 128-byte lines with it; tools/micro/handoff_litmus.hip variant 4 is
           * this mix -- write-through hand-offs, plain last stores -- with no stale read in 1e7
           * hops, and writing all 1e8 last records of csp through cost the stream kernel 3 ms) */
          store_record(h, a, t.rec_out[pid], kRecIdle);
          /* (its reach class: where the next step's pass 0 puts it inside its tile) */
          t.info_out[pid] = slot_summary(
              kRecIdle, h.cellx - a.x_off, h.celly - a.y_off, t.tiles_x, t.tile_shift,
              t.reach_classes > 1 ? reach_class(h.omega_x, h.omega_y, h.cellx - a.x_off,
                                                h.celly - a.y_off, t.tile_shift, kW, t.cells_per_x,
                                                t.cells_per_y)
                                  : 0u);
          has = false;
          ended = true;
          did_census = (h.ev == kEvCensus);
          PHASE(6);
        }
      }
      /* ---- histories handed on: the record carries the state ---- */
      /* A migrant goes to the queue of the tile it has reached, to be streamed on under a
       * window centred there by whichever workgroup claims it, inside this launch: the record
       * is stored write-through, the wave waits for its stores, reserves places in the tiles'
       * logs -- one atomic per tile the wave's migrants go to, all of them in one instruction
       * -- and writes the slots there (neutral_history.h: records that change hands inside a
       * launch).  A log that is full sends its migrant to the next pass instead. */
      bool queued = false;
      const bool hop = queues && park == kRecMigrate;
      const unsigned long long m_hop = __ballot(hop);
      if (m_hop != 0ull) { /* (wave-uniform) */
        int dest = 0;
        if (hop) {
          dest = ((h.celly - a.y_off) >> t.tile_shift) * t.tiles_x + ((h.cellx - a.x_off) >> t.tile_shift);
          if ((unsigned)dest >= (unsigned)t.ntiles || (unsigned)pid >= (unsigned)a.nparticles) {
            atomicAdd(&a.counters->aborted, 1u); /* (a cell outside the mesh: reported, never indexed with) */
            dest = 0;
          }
          store_record_through(h, a, t.rec_out[(unsigned)pid < (unsigned)a.nparticles ? pid : 0], kRecMigrate);
        }
        /* lanes bound for the same tile: the first of them reserves for all */
        int leader = -1;
        unsigned rank = 0, group = 0;
        unsigned long long todo = m_hop;
        while (todo != 0ull) {
          const int l = __builtin_ctzll(todo);
          const int tile_l = __builtin_amdgcn_readlane(dest, l);
          const unsigned long long same = __ballot(hop && dest == tile_l);
          if (hop && dest == tile_l) {
            leader = l;
            rank = (unsigned)lane_rank(same);
            group = (unsigned)__popcll(same);
          }
          todo &= ~same;
        }
        drain_stores(); /* the records are where any workgroup sees them BEFORE their slots are */
        unsigned base = 0;
        if (hop && leader == (int)(threadIdx.x & 63)) {
          base = atomicAdd(&t.queue_tail[dest], group);
        }
        base = (unsigned)__shfl((int)base, leader < 0 ? 0 : leader, 64);
        if (hop) {
          const unsigned place = base + rank;
          if (place < qcap) {
            __hip_atomic_store(t.queue_entries + ((size_t)dest * qcap + (size_t)place), (unsigned)pid,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            queued = true;
            has = false;
          }
        }
        w_hops += (unsigned)__popcll(__ballot(queued));
        w_overflows += (unsigned)__popcll(__ballot(hop && !queued));
      }
      if (park != kRecIdle && !queued) {
        if (!hop) { /* (a migrant whose queue was full has its record in place already) */
          store_record(h, a, t.rec_out[pid], park);
        }
        t.info_out[pid] = slot_summary(
            park, h.cellx - a.x_off, h.celly - a.y_off, t.tiles_x, t.tile_shift,
            (t.reach_classes > 1 && park == kRecMigrate)
                ? reach_class(h.omega_x, h.omega_y, h.cellx - a.x_off, h.celly - a.y_off,
                              t.tile_shift, kW, t.cells_per_x, t.cells_per_y)
                : 0u);
        has = false;
        ended = true;
      }
      PHASE(7);
      w_ended += (unsigned)__popcll(__ballot(ended));
      w_migrants += (unsigned)__popcll(__ballot(park == kRecMigrate && !queued));
      if (kDomain) {
        w_emigrants += (unsigned)__popcll(__ballot(park == kRecEmigrate));
      }
      w_census += (unsigned)__popcll(__ballot(did_census));
    }
  }
  PHASE(0);
  PHASE_REPORT
  /* one atomic per wave for the whole kernel: a per-event add to this single
   * word costs more than the streaming itself (one address takes ~100 adds/us) */
  if ((threadIdx.x & 63) == 0 && w_migrants) {
    atomicAdd(&t.ctrl[kCtrlMigrants], w_migrants);
  }
  if ((threadIdx.x & 63) == 0 && w_hops) {
    atomicAdd(&t.ctrl[kCtrlHops], w_hops);
  }
  if ((threadIdx.x & 63) == 0 && w_overflows) {
    atomicAdd(&t.ctrl[kCtrlOverflows], w_overflows);
  }
  if (threadIdx.x == 0 && wg_batches) {
    atomicAdd(&t.ctrl[kCtrlBatches], wg_batches);
  }
  if (threadIdx.x == 0 && wg_idle_polls) {
    atomicAdd(&t.ctrl[kCtrlIdlePolls], wg_idle_polls);
  }
  if (kDomain && (threadIdx.x & 63) == 0 && w_emigrants) {
    atomicAdd(&t.ctrl[kCtrlEmigrants], w_emigrants);
  }
  if (cur_tile >= 0) {
    flush_window<kW>(a, window, a.tally, win_ox, win_oy);
    if (kFlux) {
      flush_window<kW>(a, window + kWindowDoubles, a.flux_tally, win_ox, win_oy);
    }
  }
  {
    const unsigned wf = wave_sum_u32(nfacets);
    if ((threadIdx.x & 63) == 0) {
      if (w_processed) atomicAdd(&a.counters->nprocessed, (unsigned long long)w_processed);
      if (wf) atomicAdd(&a.counters->nfacets, (unsigned long long)wf);
      if (w_census) atomicAdd(&a.counters->ncensus, (unsigned long long)w_census);
    }
  }
  clock_stamp_end(a.counters);
}

/* ---- 2b. spatial domain decomposition: what moves between ranks ----------------------
 * A rank owns a block of the mesh (uniform blocks of bx x by cells over a px x py grid
 * of ranks) and the particles inside it.  A history that crosses out of the block is
 * stored as kRecEmigrate (stream kernel, collision stage); once the rank has nothing
 * else to do, the emigrants are packed by destination, exchanged, appended at the
 * receiver as migrants, and the step goes on there -- rounds repeat until no rank has
 * emigrants.  The reference has only the plumbing of this (x_off / y_off / neighbours,
 * neutral_interface.h:13-15; PARTICLE_SENT, neutral_data.h:35; send_and_mark_particle
 * is declared, omp3/neutral.h:63, and never defined). */

__device__ __forceinline__ int owner_rank(const DomainGrid& d, int cellx, int celly) {
  int rx = cellx / d.bx;
  int ry = celly / d.by;
  rx = (rx < 0) ? 0 : ((rx >= d.px) ? d.px - 1 : rx);
  ry = (ry < 0) ? 0 : ((ry >= d.py) ? d.py - 1 : ry);
  return ry * d.px + rx;
}

/* counts[d] = emigrants bound for rank d */
__global__ __launch_bounds__(kSortBlock) void emigrant_count_kernel(const ParticleRec* rec,
                                                                    const unsigned* info, int n,
                                                                    DomainGrid d, unsigned* counts) {
  const int i = blockIdx.x * kSortBlock + threadIdx.x;
  if (i < n && summary_state(info[i]) == kRecEmigrate) {
    atomicAdd(&counts[owner_rank(d, rec[i].cellx, rec[i].celly)], 1u);
  }
}

/* send[offset[d] + k] = the k-th emigrant bound for rank d, as a migrant; its slot is
 * marked gone.  cursor[] starts at zero. */
__global__ __launch_bounds__(kSortBlock) void emigrant_pack_kernel(ParticleRec* rec, unsigned* info,
                                                                   int n, DomainGrid d,
                                                                   const unsigned* offset,
                                                                   unsigned* cursor,
                                                                   ParticleRec* send, int tiles_x,
                                                                   int tile_shift,
                                                                   unsigned* free_slots,
                                                                   unsigned* nfree) {
  const int i = blockIdx.x * kSortBlock + threadIdx.x;
  if (i < n && summary_state(info[i]) == kRecEmigrate) {
    ParticleRec r = rec[i];
    const int dest = owner_rank(d, r.cellx, r.celly);
    const unsigned k = atomicAdd(&cursor[dest], 1u);
    free_slots[atomicAdd(nfree, 1u)] = (unsigned)i; /* an arrival can have the slot */
    const unsigned counter = record_counter(r.dead);
    r.dead = record_word(kRecMigrate, counter);
    send[offset[dest] + k] = r;
    rec[i].dead = record_word(kRecGone, 0u);
    info[i] = slot_summary(kRecGone, 0, 0, tiles_x, tile_shift);
  }
}

/* arrivals take the slots emigrants left (the last `reuse` entries of the free list),
 * then the slots behind the rank's records, and wait for the next pass */
__global__ __launch_bounds__(kSortBlock) void immigrant_append_kernel(const ParticleRec* recv,
                                                                      int nrecv, ParticleRec* rec,
                                                                      unsigned* info, unsigned* ids,
                                                                      int first_slot, int x_off,
                                                                      int y_off, int tiles_x,
                                                                      int tile_shift,
                                                                      unsigned* migrants,
                                                                      const unsigned* free_slots,
                                                                      int nfree, int reuse) {
  const int j = blockIdx.x * kSortBlock + threadIdx.x;
  if (j == 0) {
    atomicAdd(migrants, (unsigned)nrecv); /* the next pass has work: it looks at this count */
  }
  if (j < nrecv) {
    const ParticleRec r = recv[j];
    const int slot = (j < reuse) ? (int)free_slots[nfree - 1 - j] : first_slot + (j - reuse);
    rec[slot] = r;
    ids[slot] = r.id;
    info[slot] = slot_summary(kRecMigrate, r.cellx - x_off, r.celly - y_off, tiles_x, tile_shift);
  }
}

/* end of a step: the records that are still here, without the holes the emigrants
 * left (order is whatever the atomics make it; nothing depends on it) */
__global__ __launch_bounds__(kSortBlock) void compact_records_kernel(
    const ParticleRec* rec, const unsigned* info, const unsigned* ids, int n, ParticleRec* rec_to,
    unsigned* info_to, unsigned* ids_to, unsigned* cursor) {
  const int i = blockIdx.x * kSortBlock + threadIdx.x;
  const bool keep = i < n && summary_state(info[i]) != kRecGone;
  const unsigned long long m = __ballot(keep);
  unsigned base = 0;
  if ((threadIdx.x & 63) == 0 && m) {
    base = atomicAdd(cursor, (unsigned)__popcll(m));
  }
  base = __builtin_amdgcn_readfirstlane(base);
  if (keep) {
    const unsigned j = base + (unsigned)lane_rank(m);
    rec_to[j] = rec[i];
    info_to[j] = info[i];
    ids_to[j] = ids[i];
  }
}

/* decomposed stores are exchanged with the SoA arrays slot for slot (a particle's
 * index there is just where it happens to sit; its RNG key travels in keys[]) */
__global__ __launch_bounds__(kSortBlock) void import_by_slot_kernel(ParticleView p,
                                                                    const unsigned* keys,
                                                                    ParticleRec* rec, unsigned* info,
                                                                    unsigned* ids, int tiles_x,
                                                                    int tile_shift, int x_off,
                                                                    int y_off, int n) {
  const int i = blockIdx.x * kSortBlock + threadIdx.x;
  if (i < n) {
    ParticleRec r;
    r.x = p.x[i];
    r.y = p.y[i];
    r.omega_x = p.omega_x[i];
    r.omega_y = p.omega_y[i];
    r.energy = p.energy[i];
    r.weight = p.weight[i];
    r.dt_to_census = p.dt_to_census[i];
    r.mfp_to_collision = p.mfp_to_collision[i];
    r.cellx = p.cellx[i];
    r.celly = p.celly[i];
    r.id = keys[i];
    r.dead = p.dead[i] ? kRecDead : kRecIdle;
    rec[i] = r;
    ids[i] = r.id;
    info[i] = slot_summary(p.dead[i] ? kRecDead : kRecIdle, r.cellx - x_off, r.celly - y_off,
                           tiles_x, tile_shift);
  }
}

__global__ __launch_bounds__(kSortBlock) void export_by_slot_kernel(const ParticleRec* rec,
                                                                    ParticleView p, unsigned* keys,
                                                                    int n) {
  const int k = blockIdx.x * kSortBlock + threadIdx.x;
  if (k < n) {
    const ParticleRec r = rec[k];
    p.x[k] = r.x;
    p.y[k] = r.y;
    p.omega_x[k] = r.omega_x;
    p.omega_y[k] = r.omega_y;
    p.energy[k] = r.energy;
    p.weight[k] = r.weight;
    p.dt_to_census[k] = r.dt_to_census;
    p.mfp_to_collision[k] = r.mfp_to_collision;
    p.cellx[k] = r.cellx;
    p.celly[k] = r.celly;
    p.dead[k] = (record_state(r.dead) == kRecDead) ? 1 : 0;
    keys[k] = r.id;
  }
}

hipError_t launch_emigrant_count(const TiledArgs& t, int n, const DomainGrid& d, unsigned* counts,
                                 hipStream_t stream) {
  if (n > 0) {
    hipLaunchKernelGGL(emigrant_count_kernel, dim3((n + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, t.rec_out, t.info_out, n, d, counts);
  }
  return hipGetLastError();
}

hipError_t launch_emigrant_pack(const TiledArgs& t, int n, const DomainGrid& d,
                                const unsigned* offset, unsigned* cursor, ParticleRec* send,
                                unsigned* free_slots, unsigned* nfree, hipStream_t stream) {
  if (n > 0) {
    hipLaunchKernelGGL(emigrant_pack_kernel, dim3((n + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, t.rec_out, t.info_out, n, d, offset, cursor,
                       send, t.tiles_x, t.tile_shift, free_slots, nfree);
    /* the emigrants are on their way: the count starts over */
    (void)hipMemsetAsync(&t.ctrl[kCtrlEmigrants], 0, sizeof(unsigned), stream);
  }
  return hipGetLastError();
}

hipError_t launch_immigrant_append(const TiledArgs& t, const ParticleRec* recv, int nrecv,
                                   int first_slot, int x_off, int y_off,
                                   const unsigned* free_slots, int nfree, int reuse,
                                   hipStream_t stream) {
  if (nrecv > 0) {
    hipLaunchKernelGGL(immigrant_append_kernel, dim3((nrecv + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, recv, nrecv, t.rec_out, t.info_out, t.id_out,
                       first_slot, x_off, y_off, t.tiles_x, t.tile_shift, &t.ctrl[kCtrlMigrants],
                       free_slots, nfree, reuse);
  }
  return hipGetLastError();
}

hipError_t launch_compact_records(const TiledArgs& t, int n, unsigned* cursor, hipStream_t stream) {
  (void)hipMemsetAsync(cursor, 0, sizeof(unsigned), stream);
  if (n > 0) {
    hipLaunchKernelGGL(compact_records_kernel, dim3((n + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, t.rec_out, t.info_out, t.id_out, n, t.rec_in,
                       t.info_in, t.id_in, cursor);
  }
  return hipGetLastError();
}

hipError_t launch_import_by_slot(const ParticleView& p, const unsigned* keys, const TiledArgs& t,
                                 int x_off, int y_off, int n, hipStream_t stream) {
  if (n > 0) {
    hipLaunchKernelGGL(import_by_slot_kernel, dim3((n + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, p, keys, t.rec_in, t.info_in, t.id_in, t.tiles_x,
                       t.tile_shift, x_off, y_off, n);
  }
  return hipGetLastError();
}

hipError_t launch_export_by_slot(const ParticleRec* rec, const ParticleView& p, unsigned* keys,
                                 int n, hipStream_t stream) {
  if (n > 0) {
    hipLaunchKernelGGL(export_by_slot_kernel, dim3((n + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, rec, p, keys, n);
  }
  return hipGetLastError();
}

/* ---- launcher ---------------------------------------------------------------------- */

/* does the stream kernel these arguments select stage the cs index in LDS?  (the kernel's
 * kCarried, a compile-time property of the instantiation: same tables, one rank's whole mesh,
 * no tile queues -- whether or not this step's carried values are valid) */
static bool stream_stages_no_index(const SolveArgs& a, const TiledArgs& t) {
  return kCarriedStart && a.same_tables && !a.decomposed && !t.queue_entries;
}

/* edge of the stream kernel's tally window(s) in cells */
static int stream_window_cells(const SolveArgs& a, const TiledArgs& t) {
  return window_cells(a.flux_tally != nullptr, stream_stages_no_index(a, t));
}

size_t tiled_lds_bytes(const SolveArgs& a, const TiledArgs& t) {
  const int w = stream_window_cells(a, t);
  const int cells = (a.flux_tally ? 2 : 1) * w * (w + kWindowRowPad);
  const bool carried = stream_stages_no_index(a, t);
  return (carried ? stream_lds_payload_bytes<true, false>(a, cells)
                  : a.same_tables ? stream_lds_payload_bytes<true>(a, cells)
                                  : stream_lds_payload_bytes<false>(a, cells)) +
         kStreamLdsControlBytes;
}

int tiled_chunk_particles(int nparticles, int compute_units) {
  /* about six chunks per workgroup when every particle is live */
  long long c = (long long)nparticles / ((long long)compute_units * 6);
  if (c > kChunkParticlesMax) c = kChunkParticlesMax;
  if (c < kChunkParticlesMin) c = kChunkParticlesMin;
  return (int)c;
}

int tiled_tile_shift(int nx, int ny, int nparticles, bool with_flux) {
  /* Tile edge T = 16 << k cells under the fixed 128-cell window.  A tile is worth a
   * window when enough particles stream through it to pay for the window's flush; the
   * estimate is the mean particle density (particles per cell): 16-cell tiles from 8
   * per cell on (the BASELINE configurations hold 60-600), 128-cell tiles -- the
   * window itself, no margin -- below 0.5 (the reference's decks as shipped: 4000^2
   * cells, 1e6 particles, 0.06 per cell; their particles cross thousands of cells per
   * step and enter every window at an edge anyway).  NEUTRAL_TILE_CELLS overrides. */
  /* (with the scalar flux two 88-cell windows share the LDS: tiles of at most 64) */
  const int largest = with_flux ? 6 : 7;
  const char* force = getenv("NEUTRAL_TILE_CELLS");
  if (force) {
    const int cells = atoi(force);
    for (int shift = 4; shift <= largest; ++shift) {
      if (cells == (1 << shift)) {
        return shift;
      }
    }
  }
  const double density = (double)nparticles / ((double)nx * (double)ny);
  const int shift = (density >= 8.0) ? 4 : (density >= 2.0) ? 5 : (density >= 0.5) ? 6 : 7;
  return shift < largest ? shift : largest;
}

int tiled_window_min_particles(int tile_shift) {
  /* particles a chunk must hold to stream under a window: a window flush costs up
   * to 16 384 (coalesced) atomics, a particle saves one scattered atomic per facet
   * it crosses inside -- a few dozen for a 16-cell tile's short flights, a hundred
   * and more when it crosses a whole 128-cell window */
  const char* force = getenv("NEUTRAL_WINDOW_MIN_PARTICLES");
  if (force) {
    return atoi(force);
  }
  /* (128-cell tiles -- the sparsest problems -- from 64 particles on: their histories
   * cross the whole window, a hundred facets and more each) */
  return (tile_shift >= 7) ? 64 : (2048 >> (tile_shift - 4));
}

void tiled_geometry(int nx, int ny, int nparticles, int tile_shift, int* tiles_x, int* tiles_y,
                    int* max_chunks) {
  const int tile = 1 << tile_shift;
  *tiles_x = (nx + tile - 1) / tile;
  *tiles_y = (ny + tile - 1) / tile;
  /* every tile can end with one partial chunk */
  *max_chunks = (*tiles_x) * (*tiles_y) + nparticles / kChunkParticlesMin + 1;
}

hipError_t launch_import_records(const ParticleView& p, ParticleRec* rec, unsigned* info,
                                 unsigned* slot_of_id, unsigned* ids, int tiles_x, int tile_shift,
                                 int x_off, int y_off, int n, hipStream_t stream) {
  if (n > 0) {
    hipLaunchKernelGGL(import_records_kernel, dim3((n + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, p, rec, info, slot_of_id, ids, tiles_x, tile_shift,
                       x_off, y_off, n);
  }
  return hipGetLastError();
}

hipError_t launch_refresh_micro(const SolveArgs& a, const TiledArgs& t, hipStream_t stream) {
  if (t.sort_end > 0) {
    hipLaunchKernelGGL(refresh_micro_kernel, dim3((t.sort_end + kSortBlock - 1) / kSortBlock),
                       dim3(kSortBlock), 0, stream, a, t);
  }
  return hipGetLastError();
}

/* the stream kernel's instantiations that start histories from the carried values: identical
 * tables (one microscopic cross section per energy), one rank's whole mesh (a decomposed store's
 * records come and go between ranks without them), no tile queues (records that change hands
 * inside a launch) */
bool tiled_uses_carried(const SolveArgs& a, const TiledArgs& t) {
  return kCarriedStart && a.same_tables && !a.decomposed && !t.queue_entries && t.carried_in &&
         t.carried_out;
}

hipError_t launch_export_records(const ParticleRec* rec, const unsigned* slot_of_id,
                                 const ParticleView& p, int n, hipStream_t stream,
                                 const int* abort_flag, const unsigned* first_inactive,
                                 unsigned final_from, const unsigned* skip_ids, int max_blocks,
                                 bool partial_ok) {
  if (n > 0) {
    int grid = (n + kSortBlock - 1) / kSortBlock;
    grid = (max_blocks > 0 && grid > max_blocks) ? max_blocks : grid;
    hipLaunchKernelGGL(export_records_kernel, dim3(grid), dim3(kSortBlock), 0, stream, rec,
                       slot_of_id, p, n, abort_flag, first_inactive, final_from, skip_ids, partial_ok ? 1 : 0);
  }
  return hipGetLastError();
}

const unsigned* tiled_first_inactive(const TiledArgs& t) { return &t.ctrl[kCtrlFirstInactive]; }

/* one stream pass: counting sort of the records that take part, chunk list, stream kernel */
static hipError_t enqueue_stream_pass(const SolveArgs& a, TiledArgs& t, int pass, int cus,
                                      size_t lds, hipStream_t stream, hipEvent_t after_sort) {
  t.pass = pass;
  t.allow_migrate = (pass + 1 < kMaxStreamPasses) ? 1 : 0;
  /* (the graveyard beyond sort_end takes no part; a grid of one block when nothing does) */
  const int grid_n = (a.nparticles + kSortBlock - 1) / kSortBlock;
  const int grid_seg = t.sort_end > 0 ? (t.sort_end + kSortSegment - 1) / kSortSegment : 1;
  const int nbins = t.nsort + 1;
  const size_t lds_bins = (nbins <= kSortLdsBins) ? sizeof(unsigned) * (size_t)nbins : 0;
  hipLaunchKernelGGL(tile_count_kernel, dim3(grid_seg), dim3(kSortBlock), lds_bins, stream, a, t);
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, t);
  hipLaunchKernelGGL(tile_scatter_kernel, dim3(grid_seg), dim3(kSortBlock), 2 * lds_bins, stream, a,
                     t);
  hipLaunchKernelGGL(tile_chunks_kernel, dim3(1), dim3(1024), 0, stream, t);
  if (pass == 0) {
    hipLaunchKernelGGL(copy_inactive_kernel, dim3(grid_n < 8192 ? grid_n : 8192), dim3(kSortBlock), 0, stream,
                       a, t);
    if (after_sort) {
      (void)hipEventRecord(after_sort, stream);
    }
  }
  if (t.queue_entries) {
    hipLaunchKernelGGL(queue_reset_kernel, dim3(t.ntiles), dim3(kSortBlock), 0, stream, t);
  }
  /* one 1024-thread workgroup per CU (the window takes most of the LDS) */
  auto launch = [&](auto kernel) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(kernel, dim3(cus), dim3(kStreamBlock), lds, stream, a, t);
  };
  /* (scalar flux and spatial decomposition are compile-time properties of the kernel:
   * the default instantiation carries no trace of either) */
  auto launch_for = [&](auto queues_tag) {
    constexpr bool kQ = decltype(queues_tag)::value;
    switch ((a.checked ? 8 : 0) | (a.same_tables ? 4 : 0) | (a.flux_tally ? 2 : 0) |
            (a.decomposed ? 1 : 0)) {
      case 15: launch(stream_kernel<true, true, true, true, kQ>); break;
      case 14: launch(stream_kernel<true, true, false, true, kQ>); break;
      case 13: launch(stream_kernel<true, false, true, true, kQ>); break;
      case 12: launch(stream_kernel<true, false, false, true, kQ>); break;
      case 11: launch(stream_kernel<false, true, true, true, kQ>); break;
      case 10: launch(stream_kernel<false, true, false, true, kQ>); break;
      case 9: launch(stream_kernel<false, false, true, true, kQ>); break;
      case 8: launch(stream_kernel<false, false, false, true, kQ>); break;
      case 7: launch(stream_kernel<true, true, true, false, kQ>); break;
      case 6: launch(stream_kernel<true, true, false, false, kQ>); break;
      case 5: launch(stream_kernel<true, false, true, false, kQ>); break;
      case 4: launch(stream_kernel<true, false, false, false, kQ>); break;
      case 3: launch(stream_kernel<false, true, true, false, kQ>); break;
      case 2: launch(stream_kernel<false, true, false, false, kQ>); break;
      case 1: launch(stream_kernel<false, false, true, false, kQ>); break;
      default: launch(stream_kernel<false, false, false, false, kQ>); break;
    }
  };
  if (t.queue_entries) {
    launch_for(std::true_type{});
  } else {
    launch_for(std::false_type{});
  }
  return hipGetLastError();
}

hipError_t launch_solve_tiled(const SolveArgs& a, TiledArgs& t, hipStream_t stream,
                              const TiledPlan& plan, int first_pass, hipEvent_t after_sort,
                              hipEvent_t after_stream, hipEvent_t after_collect,
                              int* passes_enqueued, const SplitExport* split) {
  if (passes_enqueued) {
    *passes_enqueued = first_pass;
  }
  if (a.nparticles <= 0) {
    /* (a rank of a decomposed mesh whose block is empty right now; the caller still
     * reads the events) */
    if (after_sort && first_pass == 0) (void)hipEventRecord(after_sort, stream);
    if (after_stream) (void)hipEventRecord(after_stream, stream);
    if (after_collect) (void)hipEventRecord(after_collect, stream);
    return hipSuccess;
  }
  if (a.decomposed || t.sort_end > a.nparticles || t.mirror_end < t.sort_end) {
    t.sort_end = a.nparticles; /* (no graveyard: slots come and go within a step) */
    t.mirror_end = a.nparticles;
  }
  const int cus = a.compute_units > 0 ? a.compute_units : 256; /* (the store's LaunchTuning) */
  t.chunk_particles = tiled_chunk_particles(a.nparticles, cus);
  t.refill_min = kStreamRefillMin;
  /* (plan.stream_passes is what the step before needed, plus one) */
  t.stream_repeat = (plan.stream_passes <= 2) ? kStreamRepeatOnePass : kStreamRepeat;
  t.carried = tiled_uses_carried(a, t) ? 1 : 0;
  const size_t lds = tiled_lds_bytes(a, t);
  (void)hipFuncSetAttribute((const void*)tile_scatter_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(2 * sizeof(unsigned) * kSortLdsBins));

  if (first_pass == 0 && t.edges_computed) {
    hipLaunchKernelGGL(edges_check_kernel, dim3(1), dim3(1024), 0, stream, a, t);
  }
  if (first_pass == 0 && t.tile_uniform && a.pad == 0) {
    hipLaunchKernelGGL(tile_uniform_kernel, dim3(t.ntiles), dim3(kSortBlock), 0, stream, a, t,
                       stream_window_cells(a, t));
  } else if (first_pass == 0 && t.tile_uniform) {
    /* (a padded mesh is not checked: no window of it counts as uniform, whatever an earlier
     * step of another mesh left in the flags) */
    (void)hipMemsetAsync(t.tile_uniform, 0, (size_t)t.ntiles, stream);
  }
  /* plan.stream_passes passes, back to back: nothing here waits for the device (the
   * caller reads the migrant counter with the step's counters and comes back for
   * more if the step outran the plan) */
  int pass = first_pass;
  for (int k = 0; k < plan.stream_passes && pass < kMaxStreamPasses; ++k, ++pass) {
    const hipError_t err = enqueue_stream_pass(a, t, pass, cus, lds, stream, after_sort);
    if (err != hipSuccess) {
      return err;
    }
  }
  if (passes_enqueued) {
    *passes_enqueued = pass;
  }
  if (after_stream) {
    (void)hipEventRecord(after_stream, stream);
  }

  /* 3. the suspended histories: K2 over the collision queue; it counts its
   * events in the second StepCounters record */
  hipLaunchKernelGGL(collect_suspended_kernel,
                     dim3(t.sort_end > 0 ? (t.sort_end + kSortSegment - 1) / kSortSegment : 1),
                     dim3(kSortBlock), 0, stream, a, t, t.info_out);
  if (after_collect) {
    (void)hipEventRecord(after_collect, stream);
  }
  SolveArgs c = a;
  c.blocks_per_cu = plan.blocks_per_cu;
  /* (c.max_blocks: the store's LaunchTuning, NEUTRAL_K2_MAX_BLOCKS -- a small grid makes the
   * collision stage time-slice at particle counts a CPU oracle can follow) */
  c.counters = a.counters + 1;
  c.queue = t.collide_queue;
  c.queue_len = &t.ctrl[kCtrlCollideCount];
  c.rec = t.rec_out;
  c.slot_info = t.info_out;
  c.tiles_x = t.tiles_x;
  c.tile_shift = t.tile_shift;
  c.susp = t.susp;
  c.susp_track = t.susp_track;
  c.carried = t.carried ? t.carried_out : nullptr;
  c.steal = t.steal;
  c.emigrants = &t.ctrl[kCtrlEmigrants];
  if (t.fine_index && c.same_tables) {
    c.scatter_index = t.fine_index;
    c.scatter_index_n = t.fine_index_n;
    c.scatter_index_base = t.fine_index_base;
    c.index_shift = t.fine_index_shift;
  }
  return launch_solve(c, kVariantEventSorted, stream);
}

/* the write-back of every history that did NOT go to the collision stage, on a stream of its own
 * (lowest priority), enqueued behind the collision stage's launch: it needs the collision
 * queue's marks (after_collect) and gets the CUs that stage's waves free */
hipError_t launch_split_export(const SolveArgs& a, const TiledArgs& t, const SplitExport& split,
                               hipEvent_t after_collect) {
  (void)hipStreamWaitEvent(split.side, after_collect, 0);
  (void)launch_export_records(t.rec_out, t.slot_of_id, split.p, a.nparticles, split.side, a.abort_flag,
                              split.skip_long_dead ? &t.ctrl[kCtrlFirstInactive] : nullptr, 0xFFFFFFFFu,
                              t.susp_ids, 0, split.skip_long_dead != 0);
  (void)hipEventRecord(split.done, split.side);
  return hipGetLastError();
}


}  // namespace neutral
