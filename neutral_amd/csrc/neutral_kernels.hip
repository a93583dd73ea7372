/*
 * neutral_kernels.hip -- hand-written gfx950 kernels of the over-particle
 * transport path (the tile-sorted pipeline built on top of them lives in
 * neutral_tiled.hip).
 *
 *   K0 inject_kernel           initial particle state      (omp3/neutral.c:560-630)
 *   K1 history_kernel          one lane = one history       (omp3/neutral.c:43-206)
 *   K2 history_regroup_kernel  persistent waves, lanes regrouped by next event;
 *                              also the collision stage of the tiled pipeline
 *   build_cs_index_kernel      exponent-bucketed index over a key array
 *   tables_check_kernel        are the cs tables still what the host's cached view says?
 *   probe_*                    unit access to the building blocks (known answers)
 *
 * Common ground: one work-item per history, 64-wide wavefronts, the whole
 * history in VGPRs; event bodies shared through neutral_history.h so every
 * variant computes the same bits.  K1 launches nparticles/256 workgroups
 * (>> 256 CUs for every BASELINE configuration) and lets the dispatcher balance
 * histories that differ in length by orders of magnitude across the 8 XCDs;
 * K2 launches only as many workgroups as stay resident and feeds them from a
 * queue.  The tally mesh, density, edges and cross-section tables (<= ~11 MB
 * together) are shared by all workgroups and stay L2/Infinity-Cache resident;
 * in these two kernels tallies go to the mesh with native f64 atomics
 * (global_atomic_add_f64).  No MFMA: there is no dense contraction anywhere on
 * this path.
 */
#include "neutral_kernels.h"

#include "neutral_device.h"
#include "neutral_history.h"
#include "neutral_wave.h"

namespace neutral {

constexpr int kBlock = 256;

/* minimum resident waves per SIMD the register allocator must leave room for
 * (second __launch_bounds__ argument): 3 <=> at most 168 VGPRs, 4 <=> 128.  Variant 1
 * (K2 over the SoA store) needs ~165 without spilling and runs at 3: at 4 waves its
 * spills cost 6-20 % (profiles/r01g/baseline_configs.log).  The collision stage of the
 * tiled pipeline with identical tables and no flux tally -- the default -- needs 139
 * since the range tests that cannot fire left the event bodies, and runs at 4 with 28 B
 * of scratch in cold paths (-2.5 % against 3 waves); its other instantiations (two
 * distinct tables, scalar flux: 143-150 VGPRs) stay at 3. */
#define NEUTRAL_COLD_ARGS(a) (kQueue ? cold_args() : (a))

/* ---- K0: injection --------------------------------------------------------- */

/* Cell of coordinate c in a monotone edge array: the first ii in [0, n) with
 * edge[ii] <= c < edge[ii+1], or 0 when there is none -- what the linear scan
 * at omp3/neutral.c:590-603 returns, found by bisection. */
__device__ __forceinline__ int find_cell(const double* __restrict__ edge, int n, double c) {
  if (!(c >= edge[0]) || !(c < edge[n])) {
    return 0;
  }
  int lo = 0;
  int hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (c < edge[mid]) {
      hi = mid;
    } else {
      lo = mid;
    }
  }
  return lo;
}

__global__ __launch_bounds__(kBlock) void inject_kernel(InjectArgs a) {
  const int kk = blockIdx.x * kBlock + threadIdx.x;
  if (kk >= a.nparticles) {
    return;
  }
  const uint64_t pkey = a.pid_base + (uint64_t)kk;

  double rn0, rn1;
  generate_random_numbers(pkey, 0, 0, rn0, rn1); /* omp3/neutral.c:581 */
  const double px = a.left_off + rn0 * a.width;
  const double py = a.bottom_off + rn1 * a.height;

  const int cellx = a.x_off + find_cell(a.edgex + a.pad, a.local_nx, px);
  const int celly = a.y_off + find_cell(a.edgey + a.pad, a.local_ny, py);

  generate_random_numbers(pkey, 0, 1, rn0, rn1); /* omp3/neutral.c:611 */
  const double theta = 2.0 * M_PI * rn0;
  double s, c;
  sincos(theta, &s, &c);

  a.p.x[kk] = px;
  a.p.y[kk] = py;
  a.p.cellx[kk] = cellx;
  a.p.celly[kk] = celly;
  a.p.omega_x[kk] = c;
  a.p.omega_y[kk] = s;
  a.p.energy[kk] = a.initial_energy;
  a.p.weight[kk] = 1.0;
  a.p.dt_to_census[kk] = a.dt;
  a.p.mfp_to_collision[kk] = 0.0;
  a.p.dead[kk] = 0;
}

/* cell of c in the block's edges, or -1 when c lies outside the block */
__device__ __forceinline__ int find_cell_in_block(const double* __restrict__ edge, int n, double c) {
  if (!(c >= edge[0]) || !(c < edge[n])) {
    return -1;
  }
  return find_cell(edge, n, c);
}

/* spatial decomposition: the same particles as inject_kernel makes on one rank (same
 * streams, same arithmetic: positions in the global source box), each kept by the rank
 * whose block of the mesh it falls into */
__global__ __launch_bounds__(kBlock) void inject_filtered_kernel(InjectArgs a, unsigned* keys,
                                                                 unsigned* count) {
  const long long kk = (long long)blockIdx.x * kBlock + threadIdx.x;
  bool mine = false;
  double px = 0.0, py = 0.0;
  int cx = -1, cy = -1;
  if (kk < a.nparticles) {
    double rn0, rn1;
    generate_random_numbers((uint64_t)kk, 0, 0, rn0, rn1); /* omp3/neutral.c:581 */
    px = a.left_off + rn0 * a.width;
    py = a.bottom_off + rn1 * a.height;
    cx = find_cell_in_block(a.edgex + a.pad, a.local_nx, px);
    cy = find_cell_in_block(a.edgey + a.pad, a.local_ny, py);
    mine = (cx >= 0) & (cy >= 0);
  }
  const unsigned long long m = __ballot(mine);
  unsigned base = 0;
  if ((threadIdx.x & 63) == 0 && m) {
    base = atomicAdd(count, (unsigned)__popcll(m));
  }
  base = __builtin_amdgcn_readfirstlane(base);
  if (mine) {
    const unsigned slot = base + (unsigned)lane_rank(m);
    double rn0, rn1;
    generate_random_numbers((uint64_t)kk, 0, 1, rn0, rn1); /* omp3/neutral.c:611 */
    const double theta = 2.0 * M_PI * rn0;
    double s, c;
    sincos(theta, &s, &c);
    a.p.x[slot] = px;
    a.p.y[slot] = py;
    a.p.cellx[slot] = a.x_off + cx;
    a.p.celly[slot] = a.y_off + cy;
    a.p.omega_x[slot] = c;
    a.p.omega_y[slot] = s;
    a.p.energy[slot] = a.initial_energy;
    a.p.weight[slot] = 1.0;
    a.p.dt_to_census[slot] = a.dt;
    a.p.mfp_to_collision[slot] = 0.0;
    a.p.dead[slot] = 0;
    keys[slot] = (unsigned)kk;
  }
}

hipError_t launch_inject_filtered(const InjectArgs& a, unsigned* keys, unsigned* count,
                                  hipStream_t stream) {
  (void)hipMemsetAsync(count, 0, sizeof(unsigned), stream);
  if (a.nparticles > 0) {
    hipLaunchKernelGGL(inject_filtered_kernel, dim3((a.nparticles + kBlock - 1) / kBlock),
                       dim3(kBlock), 0, stream, a, keys, count);
  }
  return hipGetLastError();
}

/* ---- K1: over-particle history kernel -------------------------------------- */

template <bool kSameTables, bool kFlux, bool kChecked>
__global__ __launch_bounds__(kBlock, 3) void history_kernel(SolveArgs a) {
  const int pid = blockIdx.x * kBlock + threadIdx.x;

  unsigned nfacets = 0;
  unsigned ncollisions = 0;
  unsigned nprocessed = 0;
  unsigned ncensus = 0;

  if (a.abort_flag && *a.abort_flag) {
    return; /* the cached view of the cs tables is stale: the host re-runs the step */
  }
  if (pid < a.nparticles && !a.p.dead[pid]) { /* omp3/neutral.c:91-93 */
    nprocessed = 1;
    const CsLookup<const unsigned short*> ix{a.scatter_index, a.absorb_index};
    const GlobalTallyT<kFlux> tally;
    History h;
    load_particle(h, a, pid);
    prologue<kSameTables, kChecked>(h, a, ix);
    bool died = false;
    for (;;) { /* omp3/neutral.c:134-197 */
      decide(h, a);
      if (h.ev == kEvCollision) {
        ncollisions++;
        /* (a history that dies is stored where it dies: collide() has no early return) */
        if (collide<kSameTables, kChecked>(h, a, ix, tally,
                                           [&](const History& d) { store_particle(d, a, pid); })) {
          died = true;
          break;
        }
      } else if (h.ev == kEvFacet) {
        nfacets++;
        cross_facet<kChecked>(h, a, tally);
      } else {
        if (h.ev == kEvCensus) {
          ncensus = 1;
          census<kChecked>(h, a, tally);
        }
        break;
      }
    }
    if (!died) {
      store_particle(h, a, pid);
    }
  }
  flush_counters(a, nprocessed, nfacets, ncollisions, ncensus);
}

/* ---- K2: event-regrouped persistent waves ----------------------------------- */

/*
 * K1 leaves most of a wave idle whenever histories differ: in csp the few lanes
 * inside the dense block run hundreds of ~4000-cycle collisions while the other
 * lanes, done after ~60 cheap facets, wait (0.74 ns/collision against 0.027 in
 * the all-colliding scatter deck, profiles/r01/ablate_tally.log).  K2 keeps the
 * one-lane-one-history register residency but decouples lanes from particle
 * ids:
 *
 *   - persistent waves pull particle ids from a global queue in chunks
 *     (one atomicAdd per kQueueChunk ids);
 *   - every lane always knows which PASS it needs next: REFILL (no particle),
 *     STREAM (next event is a facet or the census) or COLLIDE;
 *   - each iteration the wave ballots the three populations and runs ONE pass,
 *     for the lanes that want it only; the others stay parked in registers.
 *     Expensive collisions wait until they fill most of the wave, cheap facet
 *     crossings run whenever any lane wants one, and a finished lane is re-used
 *     instead of idling until the slowest history of its wave ends.
 *
 * A lane runs exactly the event sequence K1 would run for the same particle
 * (same neutral_history.h bodies, same RNG counters), so particle end states
 * are bit-identical to K1; only the order of the tally atomics differs.
 * Exit: the queue head only grows (first-come queue) or the wave's own ring only
 * shrinks between the swaps of its own lanes (pooled mode, below); a wave leaves
 * when nothing is left to claim and none of its lanes holds a particle -- every
 * wave reaches that state without waiting for any other.
 */
constexpr int kQueueChunk = 128; /* ids a wave claims per atomic when work is plentiful */
constexpr int kRefillMin = 3;   /* REFILL pass once this many lanes are empty */
constexpr int kCollideMin = 48; /* COLLIDE pass once this many lanes wait */

enum Want : int { kWantRefill = 0, kWantStream = 1, kWantCollide = 2, kWantNothing = 3 };

/* ---- time slicing of the collision stage (queue mode) ---------------------------
 * A collider is a serial chain (csp: 931 collisions, 2.4-4.9 ms depending on how
 * many waves share its SIMD), and chains of one deck are about equally long.
 * Handing ids out first-come-first-served therefore runs in GENERATIONS: with
 * 1.5 histories per lane the second generation occupies every wave at half its
 * lanes for another full chain (9.8 ms where the work is worth 7.3), and with
 * many generations the last one still ends ragged.  So every wave takes an
 * equal, STRIDED share of the queue (wave w: entries w, w + #waves, ...; strided
 * because the queue is in tile order and chain length follows position) and
 * ROUND-ROBINS its lanes over the share: every kSlicePasses collision passes
 * the colliding lanes put their histories at the back of the wave's ring
 * (record + SuspendExtra) and take the ones at the front, as long as any are
 * waiting.  All histories of a wave then finish within one slice of each other,
 * every lane stays busy until then, and no wave depends on another (no
 * in-launch hand-off between waves, nothing to wait for).
 * The ring is the wave's own slots of the queue array: at most `share`
 * histories are ever outstanding, so position i of the ring is queue entry
 * w + (i mod share) * #waves; only this wave reads or writes those words.
 * A swap costs a record store/load and resume() -- about a fifth of a collision
 * every kSlicePasses collisions.  Histories execute exactly the events they
 * would execute unsliced: the record, the RNG counter and the pending
 * deposition are all that survives a loop head (see resume()). */
constexpr int kSlicePasses = 64;
/* Measured (profiles/r01g/ablate_shares.log, ablate_slicewindow.log):
 *  - slicing THROUGHOUT a share beats slicing only near its end by 9-17 %: lanes
 *    that swap together stay at the same collision count, hence at similar
 *    energies, and their cs-table probes fall into few cache lines; lanes refilled
 *    one by one drift apart and every probe becomes 64 separate L1 requests.
 *  - strided or contiguous shares make no difference.
 *  - with tens of generations per wave (scatter, split at 1e8: 30 000 histories per
 *    wave) the first-come-first-served queue of variant 1 is 3-6 % faster than
 *    slicing (no record round trips, and its ragged end is under 1 % of the
 *    run), so shares above kPoolMaxShare keep it. */
constexpr int kPoolMaxShare = 2048;
constexpr unsigned kRequeued = 0x80000000u; /* ring entry flag: SuspendExtra is valid */

/* ---- the waves of a CU finish together: stealing between their rings (pooled mode) ---
 * A SIMD issues for its OLDEST ready wave first.  With equal shares the oldest of a SIMD's
 * four waves takes what issue it can use (60 %) and is out after a third of the stage, the
 * second after two thirds, and the youngest ends the stage ALONE -- where a collider's
 * dependent table probes are covered by nobody: 2 335 cycles per collision pass in csp
 * against 2 077 in the decks whose waves draw from one queue and leave together
 * (profiles/r03/experiments/collision_wave_exit_times.log; s_setprio, turned in rotation or
 * steered by the waves' progress, does not change the order: same log).  So a wave that
 * has emptied its ring takes half of what WAITS in the fullest ring of a wave of its own
 * CU, as often as there is one to take from, and the sixteen waves of a CU end within a
 * slice of each other with all of them present until then.
 *   * Who shares a CU is read from the hardware (HW_ID, XCC_ID): every wave enters itself
 *     in the list of its key at the start of the launch.  Same CU on purpose: what a wave
 *     stored went through the L1 the thief reads through, and lies in the L2 both share.
 *     What correctness rests on is less than that: the key's XCC id (waves of one list run on
 *     one XCD, whose L2 is where a drained plain store is), and
 *       - every successful take is followed by an agent-scope acquire (the thief's L1 holds
 *         nothing stale of the records and ring words it is about to read: the CU bits of the
 *         key may be wrong, the take is still right);
 *       - a key that collects more than kCuWavesMax waves -- twice what a CU holds at a time:
 *         workgroups placed late enter themselves where the first ones have left (neutral_kernels.h)
 *         -- does not name one CU (the decode is wrong on this part, or the launch is not the
 *         shape the lists assume): the launch
 *         then steals nothing (StealWork::overfull; counted in StepCounters::steal_refused
 *         and NeutralHipStepStats::steals_refused);
 *       - an owner does not write into its ring while anybody reads from it (ring_readers).
 *   * A ring's control word (head << 32 | waiting) lives in memory: its owner takes from the
 *     head by compare-and-swap and appends by an atomic add (the tail -- head + waiting --
 *     moves only when the owner appends, so it is the owner's own); a thief takes from the
 *     head by compare-and-swap.  The entries a take frees lie behind the tail by as many
 *     places as histories left the ring for good, and the owner waits for the thief's copy
 *     to be over before it stores into its ring again, so nobody writes where somebody reads.
 *   * Histories are independent and carry their whole state in their record (+ SuspendExtra),
 *     so which wave finishes a history changes nothing it computes.
 *   * The words live in the tiled workspace (StealWork, neutral_kernels.h): one per workspace,
 *     reset by steal_reset_kernel on the launch's own stream -- not process-wide symbols. */
constexpr int kOldestWeight = 5; /* share of the queue a wave of the launch's first row starts with,
                                    in shares of the others (SolveArgs::share_weight) */
constexpr int kWeightedShareMin = 256; /* ... from this many histories per wave on */
constexpr int kStealMin = 96; /* waiting histories a ring must hold to be taken from
                                               (SolveArgs::steal_min; 0: no stealing) */

__device__ __forceinline__ unsigned cu_key() {
  /* HW_ID (s_getreg id 4): cu [11:8], sh [12], se [15:13]; XCC_ID (id 20): [3:0] */
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));
  return ((hw >> 8) & 0xFFu) | ((xcc & 0xFu) << 8);
}
__device__ __forceinline__ unsigned long long ring_ctl_load(StealWork* sw, int wave) {
  return __hip_atomic_load(&sw->ring_ctl[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned wave_uniform(unsigned v) {
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

/* the lists start empty, every entry invalid, nobody reading (one launch per collision stage) */
__global__ __launch_bounds__(256) void steal_reset_kernel(StealWork* sw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < kCuSlots * kCuWavesMax) {
    sw->cu_members[i] = 0xFFFFFFFFu;
  }
  if (i < kCuSlots) {
    sw->cu_count[i] = 0u;
  }
  if (i < kRingCtlSlots) {
    sw->ring_readers[i] = 0u;
    sw->ring_ctl[i] = 0ull;
  }
  if (i == 0) {
    sw->overfull = 0u;
  }
}


/* The kernel's arguments, read again from the kernarg segment: for the cold paths of the
 * collision stage (refill, hand-back, facet and census of a stray history, end of a time
 * slice).  Used there instead of `a`, their pointers and mesh constants are scalar loads at
 * the point of use instead of scalar registers held -- or, the scalar file being full,
 * spilled to vector-register lanes and read back with v_readlane -- across the collision
 * loop.  (The opaque asm keeps the loads from being hoisted back to the kernel's entry.) */
__device__ __forceinline__ SolveArgs cold_args() {
  static_assert(sizeof(SolveArgs) % 4 == 0, "SolveArgs is read back word by word");
  union {
    SolveArgs a;
    unsigned w[sizeof(SolveArgs) / 4];
  } u;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) unsigned* KernargWords;
  KernargWords p = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
#pragma unroll
  for (unsigned i = 0; i < sizeof(SolveArgs) / 4; ++i) {
    u.w[i] = p[i]; /* (scalar loads; the words nobody reads are never loaded) */
  }
#else
  __builtin_memset(&u, 0, sizeof(u));
#endif
  return u.a;
}

/* final state of a history: into the SoA store, or into its record in queue mode */
template <bool kQueue, bool kCarriesMicro = false>
__device__ __forceinline__ void put_back(const History& h, const SolveArgs& a, int pid) {
  if (kQueue) {
    const int state = h.dead ? kRecDead : kRecIdle;
    store_record(h, a, a.rec[pid], state, true); /* (a history of this stage has collided) */
    a.slot_info[pid] = slot_summary(state, h.cellx - a.x_off, h.celly - a.y_off, a.tiles_x,
                                    a.tile_shift);
    if (kCarriesMicro && a.carried) {
      /* what the next timestep's start takes from the record's slot instead of looking it up
       * (TiledArgs::carried_out): the cross section of the energy the history ends with */
      a.carried[pid].micro = h.micro_s;
    }
    if (a.export_view) {
      /* the interface's arrays get the final state here (the write-back of the histories that
       * never came to this stage runs beside it: TiledArgs::susp_ids) -- eleven scattered
       * stores per history ended, behind the stage's arithmetic: measured free, where a pass over
       * the marked ids after the stage costs 1.2 ms (profiles/r04/experiments/split_export_ab.log).
       * The eleven array pointers are fetched from memory: as kernel arguments they would be
       * live through the collision loop and push it into scratch */
      const ParticleView* pv = a.export_view;
      asm volatile("" : "+s"(pv)); /* (not foldable back into the kernel arguments) */
      store_particle_view(h, *pv, (int)h.id);
    }

  } else {
    store_particle(h, a, pid);
  }
}

/* kQueue = false: variant 1, every particle of the SoA store, streamers and
 * colliders mixed (parked lanes make occupancy matter: 4 waves/SIMD, small
 * spill); kQueue = true: the collision stage of the tiled pipeline, histories
 * suspended by the stream kernel, colliders only (3 waves/SIMD, no spill). */

template <bool kSameTables, bool kQueue, bool kFlux, bool kChecked>
__global__ __launch_bounds__(kBlock, kQueue ? ((kSameTables && !kFlux && !kChecked) ? 4 : 3)
                                             : 3)
void history_regroup_kernel(SolveArgs a) {
  unsigned nfacets = 0;
  unsigned ncollisions = 0;
  unsigned nprocessed = 0;
  unsigned ncensus = 0;

  if (a.abort_flag && *a.abort_flag) {
    return; /* the cached view of the cs tables is stale: the host re-runs the step */
  }
  if (kQueue) {
    clock_stamp_begin(cold_args().counters); /* (workgroup 0 is of the row that always works) */
  }
  /* Collision stage, launched as many workgroups per CU as stay resident (a.occupancy_rows,
   * one wave of each per SIMD): how many of them work is decided HERE, from the queue
   * length the host has not seen yet.  A collider is a serial chain of ~10^3 collisions;
   * with few of them the stage lasts one chain, and a chain runs faster the fewer waves
   * share its SIMD (csp, per chain: 2.4 ms alone, 3.6 ms with one neighbour, 4.9 ms with
   * two: profiles/r01f), so a queue that fits one (two, ...) workgroup(s) per CU keeps
   * exactly that many.  Workgroups b, b + #CUs, b + 2 #CUs, ... share a CU under every
   * dispatch order tried (and nothing breaks if they do not: this is an occupancy hint),
   * so the first `wanted` rows of #CUs workgroups stay. */
  int block_index = (int)blockIdx.x;
  int block_count = (int)gridDim.x;
  if (kQueue && a.occupancy_rows > 0) {
    const unsigned queued = *a.queue_len;
    const unsigned rows = (unsigned)a.occupancy_rows;
    const unsigned per_row = gridDim.x / rows;
    const unsigned row_lanes = per_row * (unsigned)kBlock;
    unsigned wanted = (queued + row_lanes - 1u) / row_lanes;
    wanted = (wanted < 1u) ? 1u : ((wanted > rows) ? rows : wanted);
    const unsigned row = blockIdx.x / per_row;
    if (row >= wanted) {
      return;
    }
    block_index = (int)((blockIdx.x % per_row) * wanted + row);
    block_count = (int)(per_row * wanted);
  }
  /* stage the bucketed cs index(es) in LDS: nbuckets+1 u16 entries each */
  extern __shared__ unsigned short lds_index[];
  CsLookup<const unsigned short*> ix{nullptr, nullptr};
  {
    int used = 0;
    if (a.scatter_index) {
      for (int i = threadIdx.x; i <= a.scatter_index_n; i += kBlock) {
        lds_index[i] = a.scatter_index[i];
      }
      ix.scatter_index = lds_index;
      used = a.scatter_index_n + 1;
    }
    if (!kSameTables && a.absorb_index) {
      for (int i = threadIdx.x; i <= a.absorb_index_n; i += kBlock) {
        lds_index[used + i] = a.absorb_index[i];
      }
      ix.absorb_index = lds_index + used;
    }
    __syncthreads();
  }

  const GlobalTallyT<kFlux> tally;
  /* work list: particle ids 0..nparticles-1, or the ids another kernel queued */
  const int nwork = kQueue ? (int)*a.queue_len : a.nparticles;
  /* A history is a serial chain (931 collisions of ~7 us for a csp collider), so
   * the kernel can never finish faster than the longest chain of histories one
   * wave works through.  When there are fewer ids than lanes * 2, claim them in
   * smaller chunks so that every wave gets an equal share instead of a few
   * waves getting two generations each while the other SIMDs idle. */
  const int nwaves = block_count * (kBlock / 64);
  int chunk = (nwork + nwaves - 1) / nwaves;
  chunk = (chunk < 8) ? 8 : ((chunk > kQueueChunk) ? kQueueChunk : chunk);

  History h;
  int pid = -1;
  int want = kWantRefill;
  h.ev = kEvEnd;
  /* Queue mode (colliders only, VGPRs to spare at 3 waves/SIMD): the edges of the
   * history's cell stay in registers from one collision to the next. */
  CellEdges edges{0.0, 0.0, 0.0, 0.0};
  double x_lo_open = 0.0, y_lo_open = 0.0; /* the lower edges less OPEN_BOUND_CORRECTION */
  auto next_event = [&](bool cell_changed) {
    if (kQueue) {
      if (cell_changed) {
        edges = load_edges(a, h.cellx, h.celly);
        x_lo_open = edges.x_lo - kOpenBoundCorrection;
        y_lo_open = edges.y_lo - kOpenBoundCorrection;
      }
      decide(h, a, edges);
    } else {
      decide(h, a);
    }
  };
  /* after a collision in the collision stage: the event watchdog is applied once per
   * run of back-to-back passes (below), not per collision */
  auto next_event_after_collision = [&]() {
    if (kQueue) {
      decide<false>(h, a, edges);
    } else {
      decide(h, a);
    }
  };
  unsigned long long w_ncollisions = 0; /* wave-level count (scalar) in the collision stage */

  /* wave-private slice of the particle queue (wave-uniform values) */
  int cur = 0;
  int end = 0;
  bool drained = false;

  /* pooled mode (the collision stage): the wave's strided share of the queue is its
   * ring; all wave-uniform */
  const bool pooled = kQueue && a.susp && a.steal && ((long long)nwork <= (long long)nwaves * kPoolMaxShare) &&
                      nwaves <= kRingCtlSlots;
  const int gw = block_index * (kBlock / 64) + (int)(threadIdx.x >> 6);
  /* Shares in proportion to what a wave is SERVED.  A SIMD issues for its oldest ready wave
   * first: of the four waves it holds (one of each of the CU's four workgroups: rows 0..3 of the
   * launch, row 0 dispatched first) the oldest gets ~60 % of the issue slots while all four are
   * there (7 100 collision passes against 1 500 each: collision_wave_exit_times.log of round 3).
   * With equal shares it is through long before the others -- stealing then moves work to it, but
   * only from rings of a hundred and more, and not at all at the share size an 8-GPU rank holds.
   * So the waves of row 0 start with kOldestWeight times the share of the others (5 : 1 : 1 : 1,
   * i.e. 62.5 % of a SIMD's histories), and stealing evens out what is left.  The queue is dealt
   * out to VIRTUAL waves (strided, as before), of which a wave of row 0 owns kOldestWeight
   * consecutive ones: ring position k of a wave is entry vbase + k % w + (k / w) * nvirtual.
   * Only with all four rows at work (a full launch); otherwise w = 1 and nothing changes. */
  /* (... and only where an equal share is kWeightedShareMin histories and more: below that the
   * three smaller shares no longer fill their waves' lanes -- the 8-GPU share of csp, a hundred
   * histories per wave, runs 2 % slower weighted, the full size 1 % faster:
   * profiles/r04/experiments/share_weight_ab.log) */
  const bool weighted = pooled && a.occupancy_rows == 4 && block_count == (int)gridDim.x && (nwaves % 16) == 0 &&
                        a.share_weight > 1 && (long long)nwork >= (long long)nwaves * a.weighted_share_min;
  const int oldest_weight = weighted ? a.share_weight : 1;
  const int sum_weight = oldest_weight + 3;
  const int nvirtual = weighted ? (nwaves / 4) * sum_weight : nwaves;
  struct RingMap {
    int vbase, w, share;
  };
  auto ring_map = [&](int wave) -> RingMap {
    RingMap m;
    if (weighted) {
      const int cu = wave >> 4, row = (wave >> 2) & 3, simd = wave & 3;
      m.w = (row == 0) ? oldest_weight : 1;
      m.vbase = cu * (4 * sum_weight) + simd * sum_weight + ((row == 0) ? 0 : oldest_weight + row - 1);
    } else {
      m.w = 1;
      m.vbase = wave;
    }
    /* entries of the queue this wave owns: whole rounds of its w virtual waves, and of the last,
     * partial round those that still lie inside the queue (a prefix: positions stay contiguous) */
    int n = 0;
    if (m.vbase < nwork) {
      const int rounds = (nwork - m.vbase - 1) / nvirtual; /* (full rounds before the last one) */
      const int last = nwork - m.vbase - rounds * nvirtual; /* (> 0: entries left in the last round) */
      n = rounds * m.w + ((last < m.w) ? last : m.w);
    }
    m.share = n;
    return m;
  };
  auto ring_index = [&](const RingMap& m, int pos) -> size_t {
    const int j = (m.w == 1) ? 0 : pos % m.w;
    const int round = (m.w == 1) ? pos : pos / m.w;
    return (size_t)(m.vbase + j) + (size_t)round * (size_t)nvirtual;
  };
  const RingMap my_ring = pooled ? ring_map(gw) : RingMap{0, 1, 0};
  const int share = my_ring.share;
  /* the ring's control word is a.steal->ring_ctl[gw] (StealWork); these are the owner's own view */
  unsigned w_steals = 0;
  unsigned w_steal_refused = 0;
  const unsigned w_weighted = (weighted && my_ring.share > 0) ? 1u : 0u;
  int ring_tail = 0;      /* ring position the next history handed back goes to (head + waiting) */
  int ring_count = share; /* histories waiting in the ring, as last seen (a CU-mate may have taken some) */
  int slice = 0;
  const unsigned my_cu = pooled ? cu_key() : 0u;
  if (pooled && (threadIdx.x & 63) == 0) {
    StealWork* const sw = a.steal;
    atomicExch(&sw->ring_ctl[gw], (unsigned long long)(unsigned)share); /* head 0 */
    if (a.steal_min > 0 && share > 0) {
      /* (the ring's word is in memory -- an atomic: at the L2 -- before the wave shows up in
       * its CU's list) */
      const unsigned slot = atomicAdd(&sw->cu_count[my_cu], 1u);
      if (slot < (unsigned)kCuWavesMax) {
        __hip_atomic_store(&sw->cu_members[my_cu * kCuWavesMax + slot], (unsigned)gw, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      } else {
        /* more waves under one key than a CU holds: the key does not name a CU, and what
         * stealing assumes about its members does not hold -- nobody steals in this launch */
        __hip_atomic_store(&sw->overfull, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  /* a wave whose ring is empty and whose lanes are idle: half of what waits in the fullest
   * ring of its CU, if that is worth taking (all wave-uniform) */
  auto try_steal = [&]() -> bool {
    /* (small shares -- a hundred histories per wave, the 8-GPU share of csp -- are better left
     * alone: what a thief takes there it runs in half-empty passes that the SIMD issues ahead
     * of its younger CU-mates' full ones; 48.5 against 46.5 ms per 10 steps) */
    /* (asked of the EQUAL share: with weighted shares the waves of rows 1..3 own a fifth of a
     * row-0 ring, and three quarters of the waves would never steal although the row-0 rings of
     * their CU hold hundreds) */
    const unsigned steal_min = (unsigned)a.steal_min;
    if (!pooled || steal_min == 0 || (unsigned)(nwork / nwaves) < 2u * steal_min || share == 0) {
      return false; /* (share == 0: no ring of its own to copy what it takes into) */
    }
    StealWork* const sw = cold_args().steal;
    if (__hip_atomic_load(&sw->overfull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      w_steal_refused = 1; /* (a key of more than kCuWavesMax waves: see StealWork) */
      return false;
    }
    const int lane = (int)(threadIdx.x & 63);
    /* the fullest ring among `count` waves listed at `members`: its wave, its control word */
    auto fullest = [&](const unsigned* members, unsigned count, int& v_out, unsigned& head_out) -> unsigned {
      unsigned best = 0, best_victim = 0, best_head = 0;
      for (unsigned base = 0; base < count; base += 64u) {
        unsigned victim = ~0u;
        unsigned long long ctl = 0;
        if (base + (unsigned)lane < count) {
          victim = __hip_atomic_load(&members[base + (unsigned)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          /* (the launch starts with every entry invalid; a wave that has only counted itself
           * in so far is not looked at) */
          if (victim < (unsigned)nwaves && (int)victim != gw) {
            ctl = ring_ctl_load(sw, (int)victim);
          }
        }
        const unsigned waiting = (unsigned)ctl;
        unsigned most = waiting;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned other = __shfl_xor(most, off, 64);
          most = (other > most) ? other : most;
        }
        most = wave_uniform(most);
        if (most > best) {
          const int src = __builtin_ctzll(__ballot(waiting == most));
          best = most;
          best_victim = (unsigned)__shfl(victim, src, 64);
          best_head = (unsigned)__shfl((unsigned)(ctl >> 32), src, 64);
        }
      }
      v_out = (int)best_victim;
      head_out = best_head;
      return best;
    };
    for (int attempt = 0; attempt < 4; ++attempt) {
      unsigned mates = __hip_atomic_load(&sw->cu_count[my_cu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mates = (mates > (unsigned)kCuWavesMax) ? (unsigned)kCuWavesMax : mates;
      int v = 0;
      unsigned v_head = 0;
      const unsigned best = fullest(&sw->cu_members[my_cu * kCuWavesMax], mates, v, v_head);
      if (best < steal_min || best < 2u) {
        return false;
      }
      const RingMap v_ring = ring_map(v);
      const int v_share = v_ring.share;
      /* half of it -- but a wave's worth where there is one: what a thief takes it runs in
       * passes of its own, and a half-empty pass costs the SIMD what a full one costs */
      unsigned take = best / 2u;
      take = (take < 64u) ? ((best < 64u) ? best : 64u) : take;
      take = (take > (unsigned)share) ? (unsigned)share : take;
      unsigned new_head = v_head + take;
      new_head = (new_head >= (unsigned)v_share) ? new_head - (unsigned)v_share : new_head;
      unsigned won = 0;
      if (lane == 0) {
        const unsigned long long seen = ((unsigned long long)v_head << 32) | best;
        const unsigned long long next = ((unsigned long long)new_head << 32) | (best - take);
        atomicAdd(&sw->ring_readers[v], 1u);
        won = (atomicCAS(&sw->ring_ctl[v], seen, next) == seen) ? 1u : 0u;
        if (!won) {
          atomicSub(&sw->ring_readers[v], 1u);
        }
      }
      if (!wave_uniform(won)) {
        continue; /* (its owner or another thief was quicker: look again) */
      }
      /* every take: nothing stale in this CU's L1 of what the wave reads next -- ring words and,
       * at the refill that follows, the records with plain loads -- whichever CU its owner ran
       * on (the owner's hand-back was plain stores, drained before the control word said so) */
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      /* (nobody is still reading what was taken from THIS wave's ring earlier) */
      while (__hip_atomic_load(&sw->ring_readers[gw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __builtin_amdgcn_s_sleep(2);
      }
      /* (test knob: a slow thief -- the owner of the ring it reads must wait for it) */
      for (int d = cold_args().steal_delay; d > 0; --d) {
        __builtin_amdgcn_s_sleep(64);
      }
      /* the entries [v_head, v_head + take) of v's ring are this wave's now */
      for (unsigned i = (unsigned)lane; i < take; i += 64u) {
        unsigned pos = v_head + i;
        pos = (pos >= (unsigned)v_share) ? pos - (unsigned)v_share : pos;
        const unsigned e = __hip_atomic_load(a.queue + ring_index(v_ring, (int)pos),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.queue + ring_index(my_ring, (int)i), e, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT); /* this wave's ring, from position 0 */
      }
      drain_stores(); /* (the loads have returned, the copies are where a later thief sees them) */
      if (lane == 0) {
        atomicSub(&sw->ring_readers[v], 1u);
        atomicExch(&sw->ring_ctl[gw], (unsigned long long)take); /* head 0, `take` waiting */
      }
      ring_tail = ((int)take >= share) ? (int)take - share : (int)take;
      ring_count = (int)take;
      w_steals++;
      return true;
    }
    return false;
  };
  unsigned w_requeued = 0;
  unsigned w_collide_passes = 0;
  /* ring position -> queue entry (pos < 2 * share) */
  auto ring_slot = [&](int pos) -> unsigned* {
    return a.queue + ring_index(my_ring, (pos >= share) ? pos - share : pos);
  };
  if (pooled) {
    drained = (share == 0);
  }

  for (;;) {
    const unsigned long long m_refill = __ballot(want == kWantRefill);
    const unsigned long long m_stream = __ballot(want == kWantStream);
    const unsigned long long m_collide = __ballot(want == kWantCollide);
    const int n_refill = drained ? 0 : __popcll(m_refill);
    const int n_stream = __popcll(m_stream);
    const int n_collide = __popcll(m_collide);
    if (n_refill + n_stream + n_collide == 0) {
      if (kQueue && drained && try_steal()) {
        drained = false;
        continue;
      }
      break;
    }

    /* Pass choice. */
    int pass;
    if (kQueue) {
      /* The collision stage: its histories are colliders.  The few that leak into a
       * facet crossing or reach the end of the step are served at once (a STREAM
       * pass is cheap and returns them to the collision passes, which then run
       * full: split -6...14 %); empty lanes are refilled once kRefillMin have
       * gathered (each costs a 64th of every collision pass until then: refilling
       * only when fewer than 48 lanes collide cost scatter 10 %); otherwise
       * everyone collides. */
      if (n_stream > 0) {
        pass = kWantStream;
      } else if (n_refill >= kRefillMin) {
        pass = kWantRefill;
      } else if (n_collide > 0) {
        pass = kWantCollide;
      } else {
        pass = kWantRefill;
      }
    } else if (n_refill >= kRefillMin) {
      /* variant 1, streamers and colliders mixed: collisions wait (parked in registers)
       * until they fill most of the wave; cheap STREAM passes run as long as any
       * lane wants one */
      pass = kWantRefill;
    } else if (n_collide >= kCollideMin) {
      pass = kWantCollide;
    } else if (n_stream > 0) {
      pass = kWantStream;
    } else if (n_refill > 0) {
      pass = kWantRefill;
    } else {
      pass = kWantCollide;
    }

    if (pass == kWantRefill && pooled) {
      /* ---- REFILL pass, pooled: the histories at the front of the ring ---- */
      unsigned taken = 0, head_was = 0, left = 0;
      if ((threadIdx.x & 63) == 0) {
        StealWork* const sw = cold_args().steal;
        unsigned long long seen = ring_ctl_load(sw, gw);
        for (;;) {
          const unsigned waiting = (unsigned)seen;
          head_was = (unsigned)(seen >> 32);
          taken = ((unsigned)n_refill < waiting) ? (unsigned)n_refill : waiting;
          left = waiting - taken;
          if (taken == 0) {
            break;
          }
          unsigned nh = head_was + taken;
          nh = (nh >= (unsigned)share) ? nh - (unsigned)share : nh;
          const unsigned long long next = ((unsigned long long)nh << 32) | left;
          const unsigned long long old = atomicCAS(&sw->ring_ctl[gw], seen, next);
          if (old == seen) {
            break;
          }
          seen = old; /* (a CU-mate took from the head meanwhile) */
        }
      }
      const int n_take = (int)wave_uniform(taken);
      const int ring_head = (int)wave_uniform(head_was);
      ring_count = (int)wave_uniform(left);
      const int rank = lane_rank(m_refill);
      if (want == kWantRefill && rank < n_take) {
        const SolveArgs c = NEUTRAL_COLD_ARGS(a);
        /* (ring words and records may have been written by another wave of this launch: the
         * owner's own hand-back -- through the L1 this wave reads through -- or a CU-mate's that
         * this wave took over, after the acquire its take ended with) */
        const unsigned e = __hip_atomic_load(ring_slot(ring_head + rank), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pid = (int)(e & ~kRequeued);
        SuspendExtra x;
        load_record(h, c, c.rec[pid]);
        x = c.susp[pid];
        resume<kSameTables, kChecked>(h, c, ix); /* counted as processed by the suspender */
        if (e & kRequeued) {
          h.energy_deposition = x.energy_deposition;
          h.counter = x.counter;
          h.nevents = x.nevents;
          if (kFlux) {
            h.track_length = __hip_atomic_load(&c.susp_track[pid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        next_event(true);
        want = (h.ev == kEvCollision) ? kWantCollide : kWantStream;
      }
      drained = (ring_count == 0);
    } else if (pass == kWantRefill) {
      /* ---- REFILL pass: hand fresh particle ids to the empty lanes ---- */
      if (cur >= end) {
        int base = 0;
        if ((threadIdx.x & 63) == 0) {
          base = (int)atomicAdd(&a.counters->queue_head, (unsigned)chunk);
        }
        base = __builtin_amdgcn_readfirstlane(base);
        /* the head can overshoot nparticles by at most (#waves * chunk) */
        if (base >= nwork || base < 0) {
          drained = true;
          cur = end = 0;
        } else {
          cur = base;
          end = (base + chunk < nwork) ? base + chunk : nwork;
        }
      }
      if (!drained) {
        const int mine = cur + lane_rank(m_refill);
        bool take = (want == kWantRefill) && (mine < end);
        const int avail = end - cur;
        cur += (n_refill < avail) ? n_refill : avail;
        if (kQueue) {
          /* a history the streaming kernel suspended: its record is the state */
          if (take) {
            const SolveArgs c = NEUTRAL_COLD_ARGS(a);
            pid = (int)c.queue[mine];
            load_record(h, c, c.rec[pid]);
            resume<kSameTables, kChecked>(h, c, ix); /* counted as processed by the suspender */
          }
        } else if (take && !a.p.dead[mine]) { /* omp3/neutral.c:91-93 */
          pid = mine;
          nprocessed++;
          load_particle(h, a, pid);
          prologue<kSameTables, kChecked>(h, a, ix);
        } else {
          take = false;
        }
        if (take) {
          next_event(true);
          want = (h.ev == kEvCollision) ? kWantCollide : kWantStream;
        }
      }
    } else if (pass == kWantCollide) {
      /* ---- COLLIDE pass ---- */
      /* In the collision stage the same lanes usually want the next pass as well: as
       * long as nobody's wish changes (no death, no facet, no end of step) and no
       * time slice ends, the passes follow each other without the three ballots
       * and the pass choice above. */
      bool slice_ends = false;
      const bool collides_here = (want == kWantCollide);
      unsigned inner = 0; /* passes of this run: every lane of m_collide executes all of them */
      for (;;) {
        w_collide_passes++;
        inner++;
        if (want == kWantCollide) {
          if (!kQueue) {
            ncollisions++;
          }
          if (collide<kSameTables, kChecked>(h, a, ix, tally, [&](const History& d) {
                /* (the dead start no further timestep: nothing carried for them) */
                put_back<kQueue>(d, NEUTRAL_COLD_ARGS(a), pid);
              })) {
            want = kWantRefill;
          } else if (kQueue) {
            /* (the chain goes on, or -- rarely -- its end gets a name) */
            /* Asked of the wave first in the form that needs no facet distance: where every
             * colliding lane is surely at another collision -- all but one pass in a million
             * in a dense block -- the exact comparisons are not made at all */
            bool goes_on;
            if (!kChecked && __builtin_expect(__ballot(!surely_next_is_collision(h, edges)) == 0ull, 1)) {
              goes_on = true;
            } else {
              goes_on = next_is_collision(h, edges, x_lo_open, y_lo_open);
            }
            if (goes_on) {
              h.ev = kEvCollision;
            } else {
              next_event_after_collision();
              want = (h.ev == kEvCollision) ? kWantCollide : kWantStream;
            }
          } else {
            next_event_after_collision();
            want = (h.ev == kEvCollision) ? kWantCollide : kWantStream;
          }
        }
        slice_ends = pooled && ++slice >= kSlicePasses && ring_count > 0;
        if (!kQueue || slice_ends || __ballot(want == kWantCollide) != m_collide) {
          break;
        }
      }
      if (kQueue) {
        w_ncollisions += (unsigned long long)inner * (unsigned)n_collide;
        if (collides_here) {
          h.nevents += inner;
          if (h.nevents > kMaxEventsPerHistory && want != kWantRefill) {
            atomicAdd(&a.counters->aborted, 1u);
            h.ev = kEvEnd; /* ended like a history whose time has run out */
            want = kWantStream;
          }
        }
      }
      if (slice_ends) {
        /* ---- end of a time slice: colliders swap with the waiting histories ---- */
        slice = 0;
        const bool out = (want == kWantCollide);
        const unsigned long long m_out = __ballot(out);
        StealWork* const sw = cold_args().steal;
        /* a thief may still be copying entries out of this ring (its take came first, its
         * loads may not have): nothing is stored into the ring while anybody reads from it.
         * (One load per slice, asked for here and looked at below, behind the record stores
         * -- which do not touch the ring; a copy takes microseconds.) */
        unsigned readers = __hip_atomic_load(&sw->ring_readers[gw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (out) {
          const SolveArgs c = NEUTRAL_COLD_ARGS(a);
          /* (plain stores: they are in the XCD's L2 -- the L1 writes through -- once the wave
           * has waited for them below, which is where a wave of the same XCD that takes them
           * over reads them after its acquire.  Writing them through to memory and reading them
           * around the L1 was measured too: +1.5 % on the stage, for a property the lists do not
           * need -- their key carries the XCC id) */
          store_record(h, c, c.rec[pid], kRecCollide);
          SuspendExtra x;
          x.energy_deposition = h.energy_deposition;
          x.counter = h.counter;
          x.nevents = h.nevents;
          c.susp[pid] = x;
          if (kFlux) {
            __hip_atomic_store(&c.susp_track[pid], h.track_length, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        while (wave_uniform(readers) != 0u) {
          __builtin_amdgcn_s_sleep(1);
          readers = __hip_atomic_load(&sw->ring_readers[gw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (out) {
          /* outstanding histories never exceed the share: the slot is free */
          __hip_atomic_store(ring_slot(ring_tail + lane_rank(m_out)), (unsigned)pid | kRequeued, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          want = kWantRefill;
        }
        const int n_out = __popcll(m_out);
        w_requeued += (unsigned)n_out;
        drained = false;
        /* the stores have completed, THEN the control word says that they wait */
        drain_stores();
        unsigned waiting_before = 0;
        if ((threadIdx.x & 63) == 0) {
          waiting_before = (unsigned)atomicAdd(&sw->ring_ctl[gw], (unsigned long long)n_out);
        }
        ring_count = (int)wave_uniform(waiting_before) + n_out;
        ring_tail += n_out;
        ring_tail = (ring_tail >= share) ? ring_tail - share : ring_tail;
      }
    } else {
      /* ---- STREAM pass: facet crossings, census, end of history ---- */
      if (want == kWantStream) {
        const SolveArgs c = NEUTRAL_COLD_ARGS(a);
        if (h.ev == kEvFacet) {
          nfacets++;
          cross_facet<kChecked>(h, c, tally);
          if (kQueue && c.decomposed && outside_domain(h, c)) {
            /* into another rank's cells: the history waits to be sent (its RNG counter
             * travels in the record) */
            store_record(h, c, c.rec[pid], kRecEmigrate);
            c.slot_info[pid] = slot_summary(kRecEmigrate, 0, 0, c.tiles_x, c.tile_shift);
            atomicAdd(c.emigrants, 1u);
            want = kWantRefill;
          } else {
            next_event(true);
            want = (h.ev == kEvCollision) ? kWantCollide : kWantStream;
          }
        } else {
          if (h.ev == kEvCensus) {
            ncensus++;
            census<kChecked>(h, c, tally);
          }
          put_back<kQueue, kSameTables>(h, c, pid); /* kEvEnd: the loop at :134 simply exits */
          want = kWantRefill;
        }
      }
    }
  }
  flush_counters(a, nprocessed, nfacets, ncollisions, ncensus);
  if ((threadIdx.x & 63) == 0) {
    if (w_ncollisions) atomicAdd(&a.counters->ncollisions, w_ncollisions);
    if (w_requeued) atomicAdd(&a.counters->nrequeued, (unsigned long long)w_requeued);
    if (w_steals) atomicAdd(&a.counters->nsteals, (unsigned long long)w_steals);
    if (w_steal_refused) atomicAdd(&a.counters->steal_refused, (unsigned long long)w_steal_refused);
    if (w_weighted) atomicAdd(&a.counters->nweighted, 1ull);
    if (w_collide_passes) {
      atomicAdd(&a.counters->ncollide_passes, (unsigned long long)w_collide_passes);
    }
  }
  if (kQueue) {
    clock_stamp_end(cold_args().counters);
  }
}

/* ---- bucketed cs index -------------------------------------------------------- */

__global__ __launch_bounds__(kBlock) void build_cs_index_kernel(const double* keys, int n,
                                                                int shift, long long base,
                                                                int nbuckets,
                                                                unsigned short* start) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b > nbuckets) {
    return;
  }
  /* lowest energy of bucket b; start[b] = last key <= it, clamped to [0, n-2] */
  const double e_lo = __longlong_as_double((base + (long long)b) << shift);
  int lo = 0; /* first index with keys[i] > e_lo, by bisection over [0, n] */
  int hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] > e_lo) {
      hi = mid;
    } else {
      lo = mid + 1;
    }
  }
  int s = lo - 1;
  s = (s < 0) ? 0 : ((s > n - 2) ? n - 2 : s);
  start[b] = (unsigned short)s;
}

/* ---- the cached view of the tables, re-checked every step ------------------------ */

/* The host caches what it derived from the tables (identity, bucketed indexes) and
 * this kernel re-checks the cache on the device every step, so a caller that rewrites
 * a table in place is still served correctly without a host round trip per step: on a
 * mismatch it raises out[0], the history kernels return at once, and the host
 * rebuilds its view and runs the step again.  Single workgroup. */
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

/* inside the range the fast arithmetic policy is proven on: [2^-100, 2^100], which also
 * excludes zero, negative, infinite and NaN values (neutral_device.h: the arithmetic policy) */
__device__ __forceinline__ bool is_physical(double v) { return in_proven_range(v); }

/* out[0] |= 1 if any of n values is not: the step's kernels of the fast instantiation
 * then return at entry (tables_check_kernel folds this word into the abort flag) and the
 * host runs the step with the checked one */
__global__ __launch_bounds__(1024) void unphysical_values_kernel(const double* v, long long n,
                                                                 unsigned long long* out) {
  const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  const bool odd = (i < n) && !is_physical(v[i]);
  if (__ballot(odd) != 0 && (threadIdx.x & 63) == 0) {
    atomicOr(out, 1ull);
  }
}

hipError_t launch_unphysical_values(const double* v, long long n, unsigned long long* out,
                                    hipStream_t stream) {
  if (n > 0) {
    hipLaunchKernelGGL(unphysical_values_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0,
                       stream, v, n, out);
  }
  return hipGetLastError();
}

/* (A grid of kTablesBlocks workgroups: one workgroup took 77 us for the shipped tables, on
 * the critical path of every step -- 1 % of an 8-GPU share.  Every workgroup folds its part
 * into five accumulator words with atomics and takes a ticket; the one that draws the last
 * ticket reads the totals back, writes the verdicts and clears the accumulators for the next
 * launch.) */
constexpr int kTablesBlocks = 30;

__global__ __launch_bounds__(1024) void tables_check_kernel(
    const double* ks, const double* vs, int ns, const double* ka, const double* va, int na,
    unsigned long long expect_hash_s, unsigned long long expect_hash_a, int expect_same,
    int fast_arithmetic, unsigned long long* out, unsigned long long* acc) {
  __shared__ unsigned long long s_hs[16], s_ha[16];
  __shared__ int s_diff[16];
  __shared__ int s_odd[16];
  unsigned long long hs = 0, ha = 0;
  int diff = (ns != na) ? 1 : 0;
  bool odd = false; /* a key or a value the unwrapped arithmetic is not exact on */
  const int first = (int)blockIdx.x * 1024 + (int)threadIdx.x;
  const int stride = (int)gridDim.x * 1024;
  for (int i = first; i < ns; i += stride) {
    odd = odd || !is_physical(ks[i]) || !is_physical(vs[i]);
    const unsigned long long k = (unsigned long long)__double_as_longlong(ks[i]);
    hs += mix64(k ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1)));
  }
  for (int i = first; i < na; i += stride) {
    odd = odd || !is_physical(ka[i]) || !is_physical(va[i]);
    const unsigned long long k = (unsigned long long)__double_as_longlong(ka[i]);
    ha += mix64(k ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1)));
  }
  if (ns == na) {
    for (int i = first; i < ns; i += stride) {
      /* bit comparison: NaNs and signed zeros must not compare "equal enough" */
      diff |= (__double_as_longlong(ks[i]) != __double_as_longlong(ka[i])) ||
              (__double_as_longlong(vs[i]) != __double_as_longlong(va[i]));
    }
  }
  const int wave_odd = (__ballot(odd) != 0) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) {
    hs += __shfl_down(hs, off, 64);
    ha += __shfl_down(ha, off, 64);
    diff |= __shfl_down(diff, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_hs[threadIdx.x >> 6] = hs;
    s_ha[threadIdx.x >> 6] = ha;
    s_diff[threadIdx.x >> 6] = diff;
    s_odd[threadIdx.x >> 6] = wave_odd;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    hs = ha = 0;
    diff = 0;
    int any_odd = 0;
    for (int w = 0; w < 16; ++w) {
      hs += s_hs[w];
      ha += s_ha[w];
      diff |= s_diff[w];
      any_odd |= s_odd[w];
    }
    atomicAdd(&acc[0], hs);
    atomicAdd(&acc[1], ha);
    if (diff) atomicOr(&acc[2], 1ull);
    if (any_odd) atomicOr(&acc[3], 1ull);
    __threadfence();
    const unsigned long long ticket = atomicAdd(&acc[4], 1ull);
    if (ticket != (unsigned long long)gridDim.x - 1ull) {
      return;
    }
    /* the last workgroup: everybody's contribution is in (read back through the atomics'
     * own path), and the accumulators start the next launch from zero */
    __threadfence();
    hs = atomicExch(&acc[0], 0ull);
    ha = atomicExch(&acc[1], 0ull);
    diff = (int)atomicExch(&acc[2], 0ull);
    any_odd = (int)atomicExch(&acc[3], 0ull);
    atomicExch(&acc[4], 0ull);
    const int same = diff ? 0 : 1;
    out[1] = hs;
    out[2] = ha;
    out[3] = (unsigned long long)same;
    out[4] = (unsigned long long)any_odd;
    /* why the step's history kernels must not run as launched: [6] the host's view of
     * the tables is stale; [7] the fast arithmetic was launched on input outside its
     * proven range (out[5]: the densities, by unphysical_values_kernel earlier on this
     * stream).  [0] is the abort flag the kernels read. */
    const unsigned long long stale =
        (hs != expect_hash_s || ha != expect_hash_a || same != expect_same) ? 1ull : 0ull;
    const unsigned long long unproven = (fast_arithmetic && (any_odd || out[5] != 0)) ? 1ull : 0ull;
    out[6] = stale;
    out[7] = unproven;
    out[0] = stale | unproven;
  }
}

hipError_t launch_tables_check(const double* ks, const double* vs, int ns, const double* ka,
                               const double* va, int na, unsigned long long expect_hash_s,
                               unsigned long long expect_hash_a, int expect_same,
                               int fast_arithmetic, unsigned long long* out4,
                               hipStream_t stream) {
  /* (out4 + 8 ... out4 + 12: the accumulators, zero between launches) */
  hipLaunchKernelGGL(tables_check_kernel, dim3(kTablesBlocks), dim3(1024), 0, stream, ks, vs, ns, ka,
                     va, na, expect_hash_s, expect_hash_a, expect_same, fast_arithmetic, out4,
                     out4 + 8);
  return hipGetLastError();
}

/* ---- probes: unit-level access to the device building blocks (for KATs) ----- */

__global__ __launch_bounds__(kBlock) void probe_threefry_kernel(const uint64_t* in, uint64_t* out,
                                                                 double* rn, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* in: {counter, pkey, master_key} per row */
    uint64_t r0, r1;
    threefry2x64_20(in[3 * i], in[3 * i + 1], in[3 * i + 2], r0, r1);
    out[2 * i] = r0;
    out[2 * i + 1] = r1;
    double a, b;
    generate_random_numbers(in[3 * i + 1], in[3 * i + 2], in[3 * i], a, b);
    rn[2 * i] = a;
    rn[2 * i + 1] = b;
  }
}

__global__ __launch_bounds__(kBlock) void probe_cs_kernel(const double* keys, const double* values,
                                                           int nentries, const double* energy,
                                                           double* value, int* index, int n,
                                                           const unsigned short* index_start,
                                                           int index_n, int index_shift,
                                                           long long index_base) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* index_start == null probes the plain bisection */
    const int ind = index_start ? cs_bracket_indexed(keys, nentries, index_start, index_n,
                                                     index_shift, index_base, energy[i])
                                : cs_bracket(keys, nentries, energy[i]);
    index[i] = ind;
    value[i] = cs_interpolate<false>(keys, values, ind, energy[i]);
  }
}

__global__ __launch_bounds__(kBlock) void probe_facet_kernel(const double* in, double* dist,
                                                              int* x_facet, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* in: {x, y, omega_x, omega_y, speed, ex_lo, ex_hi, ey_lo, ey_hi} per row */
    const double* r = in + 9 * i;
    double d;
    int xf;
    calc_distance_to_facet(r[0], r[1], r[2], r[3], r[4], 1.0 / (r[2] * r[4]), 1.0 / (r[3] * r[4]),
                           r[5], r[6], r[7], r[8], d, xf);
    dist[i] = d;
    x_facet[i] = xf;
  }
}

__global__ __launch_bounds__(kBlock) void probe_division_kernel(const double* in, double* out,
                                                                 int* plain, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* in: {a, b} per row; out: {a / b, quotient through the kept reciprocal} */
    const double a = in[2 * i];
    const double b = in[2 * i + 1];
    out[2 * i] = a / b;
    out[2 * i + 1] = quotient_by_reciprocal(a, b, refined_reciprocal(b));
    plain[i] = (in_plain_division_range(a) && in_plain_division_range(b)) ? 1 : 0;
  }
}

__global__ __launch_bounds__(kBlock) void probe_log_kernel(const double* in, double* out, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    out[4 * i] = log_of_sample(in[i]); /* what the history kernels use */
    out[4 * i + 1] = log(in[i]);       /* the device library's */
    out[4 * i + 2] = sqrt_plain_range(in[i]);
    out[4 * i + 3] = sqrt(in[i]);
    /* the two constant denominators of the collision, both ways */
    out[4 * n + 4 * i] = quotient_by_constant<ByParticleMass>(in[i], kParticleMass,
                                                              1.0 / kParticleMass);
    out[4 * n + 4 * i + 1] = in[i] / kParticleMass;
    out[4 * n + 4 * i + 2] = quotient_by_constant<ByMassNoPlusOneSquared>(
        in[i], kMassNoPlusOneSquared, 1.0 / kMassNoPlusOneSquared);
    out[4 * n + 4 * i + 3] = in[i] / kMassNoPlusOneSquared;
  }
}

__global__ __launch_bounds__(kBlock) void probe_scatter_kernel(const double* in, double* out, int n) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    /* in: {energy, mu_cm, omega_x, omega_y} per row */
    const double e = in[4 * i], mu_cm = in[4 * i + 1];
    const double e_new = quotient_of_physical_by_constant<ByMassNoPlusOneSquared, false>(
        e * (kMassNo * kMassNo + 2.0 * kMassNo * mu_cm + 1.0), kMassNoPlusOneSquared, 1.0 / kMassNoPlusOneSquared);
    double root_ratio, inv_root_ratio, unused0, unused1;
    const double speed = speed_of<true>(e);
    History h;
    h.omega_x = in[4 * i + 2];
    h.omega_y = in[4 * i + 3];
    h.speed = speed_of<true>(e_new);
    double* const o = out + 10 * i;
    o[0] = e_new;
    o[1] = scatter_cosine<false>(e, e_new, root_ratio, inv_root_ratio);
    o[2] = scatter_cosine<true>(e, e_new, unused0, unused1);
    o[3] = speed_after_scatter(e_new, speed, refined_reciprocal(speed), root_ratio, inv_root_ratio);
    o[4] = h.speed;
    refresh_direction_plain_or_wrapped(h);
    o[5] = h.u_x_inv;
    o[6] = h.u_y_inv;
    refresh_direction(h);
    o[7] = h.u_x_inv;
    o[8] = h.u_y_inv;
    o[9] = 0.0;
  }
}

hipError_t launch_probe_scatter(const double* in, double* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(probe_scatter_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, in, out, n);
  return hipGetLastError();
}

hipError_t launch_probe_log(const double* in, double* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(probe_log_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                     in, out, n);
  return hipGetLastError();
}

hipError_t launch_probe_division(const double* in, double* out, int* plain, int n,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(probe_division_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     stream, in, out, plain, n);
  return hipGetLastError();
}

/* ---- launchers -------------------------------------------------------------- */

LaunchTuning launch_tuning_from_env() {
  LaunchTuning t;
  auto env_int = [](const char* name, int fallback, int lo, int hi) {
    const char* v = getenv(name);
    if (v && *v) {
      const int i = atoi(v);
      if (i >= lo && i <= hi) {
        return i;
      }
    }
    return fallback;
  };
  t.steal_min = env_int("NEUTRAL_STEAL_MIN", kStealMin, 0, 1 << 20);       /* (tests: small rings taken from too) */
  t.share_weight = env_int("NEUTRAL_SHARE_WEIGHT", kOldestWeight, 1, 16);   /* (A/B and tests: 1 = equal shares) */
  t.weighted_share_min = env_int("NEUTRAL_WEIGHTED_SHARE_MIN", kWeightedShareMin, 1, 1 << 20); /* (tests) */
  t.steal_delay = env_int("NEUTRAL_STEAL_DELAY", 0, 0, 1 << 20);            /* (tests: a slow thief) */
  /* (tests: a small grid makes the collision stage time-slice -- shares larger than a wave --
   * at particle counts a CPU oracle can follow) */
  t.max_blocks = env_int("NEUTRAL_K2_MAX_BLOCKS", 0, 0, 1 << 20);
  int dev = 0;
  t.compute_units = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    (void)hipDeviceGetAttribute(&t.compute_units, hipDeviceAttributeMultiprocessorCount, dev);
  }
  return t;
}

hipError_t launch_probe_threefry(const uint64_t* in, uint64_t* out, double* rn, int n,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(probe_threefry_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     stream, in, out, rn, n);
  return hipGetLastError();
}

hipError_t launch_probe_cs(const double* keys, const double* values, int nentries,
                           const double* energy, double* value, int* index, int n,
                           const CsIndex& ix, hipStream_t stream) {
  hipLaunchKernelGGL(probe_cs_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                     keys, values, nentries, energy, value, index, n, ix.start, ix.nbuckets,
                     ix.shift, ix.base);
  return hipGetLastError();
}

hipError_t launch_build_cs_index(const double* keys, int n, int shift, long long base,
                                 int nbuckets, unsigned short* start, hipStream_t stream) {
  hipLaunchKernelGGL(build_cs_index_kernel, dim3((nbuckets + 1 + kBlock - 1) / kBlock),
                     dim3(kBlock), 0, stream, keys, n, shift, base, nbuckets, start);
  return hipGetLastError();
}

hipError_t launch_probe_facet(const double* in, double* dist, int* x_facet, int n,
                              hipStream_t stream) {
  hipLaunchKernelGGL(probe_facet_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                     in, dist, x_facet, n);
  return hipGetLastError();
}

hipError_t launch_inject(const InjectArgs& a, hipStream_t stream) {
  if (a.nparticles <= 0) {
    return hipSuccess;
  }
  const int grid = (a.nparticles + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(inject_kernel, dim3(grid), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

/* blocks of kBlock threads the device keeps resident for a kernel that takes `lds`
 * bytes of dynamic LDS per block */
template <typename K>
static int resident_blocks(K kernel, size_t lds, int compute_units) {
  const int cus = compute_units > 0 ? compute_units : 256;
  int per_cu = 2;
  /* registers decide (3 waves per SIMD: section above); the occupancy query is asked
   * without the dynamic LDS, which it prices against 64 KB per CU where gfx950 has
   * 160 KB, and the LDS bound is applied by hand */
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess ||
      per_cu < 1) {
    per_cu = 2;
  }
  if (lds > 0) {
    const int by_lds = (int)((size_t)(160 * 1024) / lds);
    per_cu = (by_lds < per_cu) ? by_lds : per_cu;
    per_cu = (per_cu < 1) ? 1 : per_cu;
  }
  return cus * per_cu;
}

hipError_t launch_solve(const SolveArgs& a, int variant, hipStream_t stream) {
  if (a.nparticles <= 0) {
    return hipSuccess;
  }
  if (variant == kVariantEventSorted || a.queue) {
    /* persistent waves: as many workgroups as stay resident, never more than
     * there are chunks of work; no workgroup depends on another, so an
     * over-estimate only queues the surplus */
    /* with a device-side queue the length is not known on the host: size for
     * the worst case, surplus waves drain at their first refill */
    const int chunks = (a.nparticles + kQueueChunk - 1) / kQueueChunk;
    const int want_blocks = (chunks + (kBlock / 64) - 1) / (kBlock / 64);
    size_t idx_entries = a.scatter_index ? (size_t)(a.scatter_index_n + 1) : 0;
    if (!a.same_tables && a.absorb_index) {
      idx_entries += (size_t)(a.absorb_index_n + 1);
    }
    const size_t lds = sizeof(unsigned short) * idx_entries;
    if (lds > (size_t)(160 * 1024 - 64)) {
      return hipErrorInvalidValue; /* the ABI drops an index before this can happen */
    }
    SolveArgs k = a;
    auto launch = [&](auto kernel) {
      /* (the indexes of two distinct large tables can exceed the 64 KB default) */
      (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
      int grid = resident_blocks(kernel, lds, a.compute_units);
      k.occupancy_rows = 0;
      const int cus = a.compute_units > 0 ? a.compute_units : 256;
      if (a.blocks_per_cu == -1) {
        /* the kernel picks its own occupancy among the rows of #CUs workgroups that are
         * resident together; otherwise everybody works */
        const int rows = grid / cus;
        const bool fits = (rows >= 2) && (grid == rows * cus) && (want_blocks >= grid) &&
                          !(a.max_blocks > 0 && a.max_blocks < grid);
        k.occupancy_rows = fits ? rows : 0;
        k.blocks_per_cu = 0;
      }
      if (a.blocks_per_cu > 0 && a.blocks_per_cu * cus < grid) {
        /* the caller knows how little work there is: fewer resident waves per
         * SIMD shorten every history's serial chain (see launch_solve_tiled) */
        grid = a.blocks_per_cu * cus;
      }
      if (grid > want_blocks) {
        grid = want_blocks;
      }
      if (a.max_blocks > 0 && grid > a.max_blocks) {
        grid = a.max_blocks;
      }
      /* (k.steal_min, share_weight, weighted_share_min, steal_delay: the store's LaunchTuning,
       * filled in by the caller -- read from the environment once per store, not here) */
      if (a.queue && k.steal) {
        /* rings' control words zero, CU lists empty (every entry invalid), nobody reading */
        hipLaunchKernelGGL(steal_reset_kernel, dim3((kCuSlots * kCuWavesMax + 255) / 256), dim3(256), 0,
                           stream, k.steal);
      } else {
        k.steal = nullptr; /* (no workspace: equal shares without rings, first-come queue) */
        k.steal_min = 0;
      }
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, k);
    };
    /* (the scalar-flux tally is a compile-time property of a kernel: the default
     * instantiations carry no trace of it) */
    /* (and so is the arithmetic policy, a.checked: neutral_device.h) */
    const int pick = (a.checked ? 8 : 0) | (a.queue ? 4 : 0) | (a.same_tables ? 2 : 0) |
                     (a.flux_tally ? 1 : 0);
    switch (pick) {
      case 15: launch(history_regroup_kernel<true, true, true, true>); break;
      case 14: launch(history_regroup_kernel<true, true, false, true>); break;
      case 13: launch(history_regroup_kernel<false, true, true, true>); break;
      case 12: launch(history_regroup_kernel<false, true, false, true>); break;
      case 11: launch(history_regroup_kernel<true, false, true, true>); break;
      case 10: launch(history_regroup_kernel<true, false, false, true>); break;
      case 9: launch(history_regroup_kernel<false, false, true, true>); break;
      case 8: launch(history_regroup_kernel<false, false, false, true>); break;
      case 7: launch(history_regroup_kernel<true, true, true, false>); break;
      case 6: launch(history_regroup_kernel<true, true, false, false>); break;
      case 5: launch(history_regroup_kernel<false, true, true, false>); break;
      case 4: launch(history_regroup_kernel<false, true, false, false>); break;
      case 3: launch(history_regroup_kernel<true, false, true, false>); break;
      case 2: launch(history_regroup_kernel<true, false, false, false>); break;
      case 1: launch(history_regroup_kernel<false, false, true, false>); break;
      default: launch(history_regroup_kernel<false, false, false, false>); break;
    }
    return hipGetLastError();
  }
  const int grid = (a.nparticles + kBlock - 1) / kBlock;
  auto launch1 = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, a);
  };
  switch ((a.checked ? 4 : 0) | (a.same_tables ? 2 : 0) | (a.flux_tally ? 1 : 0)) {
    case 7: launch1(history_kernel<true, true, true>); break;
    case 6: launch1(history_kernel<true, false, true>); break;
    case 5: launch1(history_kernel<false, true, true>); break;
    case 4: launch1(history_kernel<false, false, true>); break;
    case 3: launch1(history_kernel<true, true, false>); break;
    case 2: launch1(history_kernel<true, false, false>); break;
    case 1: launch1(history_kernel<false, true, false>); break;
    default: launch1(history_kernel<false, false, false>); break;
  }
  return hipGetLastError();
}

}  // namespace neutral
