/*
 * neutral_kernels.h -- launch-side view of the HIP kernels (host code of the
 * C-ABI includes this; kernels are defined in neutral_kernels.hip).
 */
#ifndef NEUTRAL_AMD_KERNELS_H
#define NEUTRAL_AMD_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace neutral {

/* SoA particle store in HBM: the reference's -DSoA Particle
 * (neutral_data.h:45-61).  One f64/i32 stream per field, lane i <-> particle
 * base+i, so every prologue/epilogue access is a fully coalesced 512-B (f64) or
 * 256-B (i32) wave transaction. */
struct ParticleView {
  double* x;
  double* y;
  double* omega_x;
  double* omega_y;
  double* energy;
  double* weight;
  double* dt_to_census;
  double* mfp_to_collision;
  int* cellx;
  int* celly;
  int* dead;
};

/* One particle as a record: the private working layout of the tiled variant.
 * Sorted-by-tile access is random per particle, and a random 80-B record costs
 * two 64-B sectors where the SoA store costs eleven; the records are kept in
 * tile order from step to step and carry the particle id (the RNG key). */
/* What a history that neither collides nor reflects changes comes FIRST (48 bytes: position, the
 * two clocks, cell, id, state): the write-back of such a history reads those three quads -- one
 * 64-byte sector for every other record instead of two -- and stores six fields, not eleven
 * (export_records_kernel; the state word says whether direction, energy or weight changed). */
struct alignas(16) ParticleRec {
  double x, y, dt_to_census, mfp_to_collision;
  int cellx, celly;
  unsigned id; /* index in the SoA store = global id - pid_base */
  int dead;    /* the record's state word: neutral_history.h, record_word() */
  double omega_x, omega_y, energy, weight;
};
constexpr int kParticleRecBytes = 80;
static_assert(sizeof(ParticleRec) == kParticleRecBytes, "ParticleRec holds 80 bytes");

/* What a history suspended MID-CHAIN by the collision stage's time slicing needs
 * beyond its record: the RNG counter, the deposition not yet tallied
 * (omp3/neutral.c:236: flushed only at a facet, the census or death) and the
 * watchdog count.  Indexed like the record array. */
struct alignas(16) SuspendExtra {
  double energy_deposition;
  unsigned counter;
  unsigned nevents;
};
/* (with the scalar-flux tally the pending weight * path length of such a history
 * lives in a parallel array of doubles: TiledArgs::susp_track) */

/* what the start of a history does not depend on (TiledArgs::carried_in) */
struct alignas(16) CarriedStart {
  double micro;          /* microscopic cross section of the record's energy (identical tables) */
  double minus_log_rn0;  /* -log of the timestep's first sample (omp3/neutral.c:127-131) */
};

struct InjectArgs {
  int nparticles;
  uint64_t pid_base;
  int local_nx;
  int local_ny;
  int pad;
  int x_off;
  int y_off;
  double left_off;
  double bottom_off;
  double width;
  double height;
  double dt;
  double initial_energy;
  const double* edgex;
  const double* edgey;
  ParticleView p;
};

/* device-resident event counters of one solve step */
struct StepCounters {
  unsigned long long nprocessed;
  unsigned long long nfacets;
  unsigned long long ncollisions;
  unsigned long long ncensus; /* histories that ended in a census event */
  unsigned int queue_head;    /* K2: next unclaimed particle index */
  unsigned int aborted;       /* histories stopped by the event watchdog (should be 0) */
  unsigned long long nrequeued; /* time-slice swaps of the collision stage (queue mode) */
  unsigned long long ncollide_passes; /* wave-level COLLIDE passes of the regroup kernel */
  unsigned long long nsteals; /* collision stage: rings a wave took half the waiting histories of */
  unsigned long long steal_refused; /* collision stage: waves that found a CU list of more than
                                       kCuWavesMax entries and therefore stole nothing (the key
                                       read from the hardware does not name one CU: see StealWork) */
  unsigned long long nweighted; /* collision stage: waves that were dealt a share in proportion to
                                   what a wave is served (SolveArgs::share_weight) */
  /* The shader clock the kernel ran at, measured by one wave of every launch over its own
   * life: ticks of the shader clock (s_memtime) and of the constant 100-MHz clock
   * (s_memrealtime), summed over the step's launches.  The chip does not hold its nominal
   * 2.4 GHz under every load (a kernel of nothing but v_fma_f64 runs at 1.85 GHz:
   * profiles/r05/pmc_calibration.log), and a roofline in issue cycles needs the clock there was. */
  unsigned long long clock_shader_ticks;
  unsigned long long clock_100mhz_ticks;
  unsigned long long clock_parked[2]; /* (the measuring wave's counters at its start) */
};

/* Device workspace of the collision stage's work stealing (neutral_kernels.hip): the control
 * word of every wave's ring, who is reading from which ring, and the list of waves per CU key.
 * It belongs to the tiled workspace (TiledArgs::steal), is handed to the kernel in SolveArgs and
 * is reset by one kernel on the launch's own stream before every collision stage. */
constexpr int kRingCtlSlots = 8192;  /* waves of a launch (4 096 on an MI355X) */
constexpr int kCuSlots = 4096;       /* (xcc 4 bits, se 3, sh 1, cu 4) */
/* (16 waves of a launch are resident on a CU at a time, but a key can collect more over a launch's
 * life: the write-back pass that starts beside the collision stage may hold wave slots when the
 * stage's workgroups are placed, the ones left over are placed when a slot frees up -- also on a
 * CU whose first workgroups have gone through their small shares and left, where they are the
 * fifth to enter themselves.  Twice a CU's waves: a key that collects more does not name a CU) */
constexpr int kCuWavesMax = 32;
struct StealWork {
  unsigned long long ring_ctl[kRingCtlSlots]; /* head << 32 | waiting */
  /* waves that are reading entries out of a ring they have just taken from: the owner waits
   * for zero before it writes into its ring (hand-back) or re-uses it for what it steals */
  unsigned ring_readers[kRingCtlSlots];
  unsigned cu_count[kCuSlots];
  unsigned cu_members[kCuSlots * kCuWavesMax];
  unsigned overfull; /* a CU key collected more than kCuWavesMax waves: nobody steals */
};

struct SolveArgs {
  int nx;
  int ny;
  int global_nx;
  int global_ny;
  uint64_t master_key;
  int pad;
  int x_off;
  int y_off;
  double dt;
  double inv_ntotal_particles;
  int nparticles;
  uint64_t pid_base;
  ParticleView p;
  const double* density;
  const double* edgex;
  const double* edgey;
  /* The mesh as a formula, when it is one: edge[pad + i] == edge_d * (double)(off + i), bit
   * for bit, for every edge of both axes (the way the reference's host layer makes uniform
   * meshes: edgedx[i] * (x_off + i - pad)).  The spacings are read once per mesh from the
   * caller's edgedx / edgedy (0: none given); whether the formula HOLDS is checked on the
   * device every step (TiledArgs::edges_computed), and the stream kernel then works the two
   * edges of a crossing out instead of loading them. */
  double edge_dx;
  double edge_dy;
  const double* scatter_keys;
  const double* scatter_values;
  int scatter_n;
  const double* absorb_keys;
  const double* absorb_values;
  int absorb_n;
  int same_tables; /* both tables have identical contents: search once */
  int checked;     /* arithmetic policy of the event bodies (neutral_device.h): 0 the fast
                      sequences, proven on densities and table entries in [2^-100, 2^100];
                      1 IEEE operations with range tests, any input the reference accepts */
  /* exponent-bucketed indexes over the key arrays (start == null: none) */
  const unsigned short* scatter_index;
  int scatter_index_n;
  long long scatter_index_base;
  const unsigned short* absorb_index;
  int absorb_index_n;
  long long absorb_index_base;
  int index_shift;
  double* tally;
  double* flux_tally; /* scalar-flux tally (null: not kept; see neutral_hip.h) */
  StepCounters* counters;
  /* optional work list for the regroup kernel: ids of particles another kernel
   * suspended at their first collision (null: all particles 0..nparticles-1) */
  unsigned* queue;           /* (the collision stage re-uses its words as per-wave rings) */
  const unsigned* queue_len; /* [device] number of valid entries */
  ParticleRec* rec;          /* queue entries index this record array (tiled variant) */
  int blocks_per_cu;         /* > 0: cap on the regroup kernel's workgroups per CU; -1: the
                                collision stage picks 1..3 itself from its queue length */
  int max_blocks;            /* > 0: cap on the regroup kernel's grid (test knob) */
  unsigned* slot_info;       /* per-record summary kept next to rec (see TiledArgs) */
  int tiles_x;               /* tiles per mesh row, for the summary's tile field */
  int tile_shift;            /* log2 of the tile edge in cells (4..7) */
  SuspendExtra* susp;        /* per-record side store of time-sliced histories (queue mode) */
  double* susp_track;        /* ... their pending weight * path length (scalar flux only) */
  CarriedStart* carried;     /* per-record microscopic cross section the collision stage leaves for
                                the next timestep's start (TiledArgs::carried_out; null: not kept) */
  int steal_min;              /* collision stage: waiting histories a ring must hold to be taken
                                 from by a CU-mate (0: no stealing; NEUTRAL_STEAL_MIN) */
  StealWork* steal;           /* [device] rings' control words and CU lists (null: no stealing) */
  int steal_delay;            /* test knob: sleeps of a thief between its take and its copy
                                 (NEUTRAL_STEAL_DELAY; 0 in production) */
  int occupancy_rows;         /* collision stage: workgroups per CU resident together, among
                                 which the kernel picks how many work (0: all) */
  int share_weight;           /* collision stage: how many times the others' share of the queue
                                 the waves of the launch's first row of workgroups start with (the
                                 oldest wave of a SIMD is served first; 1: equal shares) */
  int weighted_share_min;     /* ... from this many histories per wave on (equal shares below) */
  int compute_units;          /* CUs of the device the launch goes to (read once per store) */
  int export_skip_long_dead;  /* the arrays were current as the step began: particles dead
                                 since before it are left alone by the write-back pass */
  /* [device] the SoA store's array pointers, when the collision stage writes the final state
   * of every history it ends to the arrays itself -- so that the write-back of everybody else
   * can run beside it (TiledArgs::susp_ids; null: it does not).  Read where a history ends,
   * not held in registers through the collision loop. */
  const ParticleView* export_view;
  /* spatial domain decomposition: this rank owns cells [x_off, x_off + nx) x [y_off,
   * y_off + ny) of a larger mesh; a history that crosses out of them is stored as an
   * emigrant instead of going on (0: the rank owns the whole mesh) */
  int decomposed;
  unsigned* emigrants; /* [device] their count (collision stage; the stream kernel has t.ctrl) */
  /* [device] non-zero: the host's cached view of the cs tables (identity, bucketed
   * indexes) no longer matches the tables; history kernels return at once and the
   * host re-runs the step with a fresh view (null: no cached view in use) */
  const int* abort_flag;
};

/* device workspace of the tiled pipeline (neutral_tiled.hip), owned by the ABI */
struct TiledArgs {
  ParticleRec* rec_in;     /* nparticles: last step's records (source of pass 0) */
  ParticleRec* rec_out;    /* nparticles: this step's records, in this step's tile order */
  /* 4-B summary of each record, written with it: state << 29 | tile of its cell.
   * The counting sort and the collision queue are built from these 4 bytes instead
   * of a strided read of the 80-B records. */
  unsigned* info_in;
  unsigned* info_out;
  unsigned* id_in;         /* nparticles: particle id of each record (= rec.id, kept apart so */
  unsigned* id_out;        /* that a reader takes 4 bytes per record, not a 64-byte sector) */
  unsigned* slot_of_id;    /* nparticles: where particle id's record is -- what the write-back
                              goes by */
  /* What a history's start does not depend on, kept per record slot so that the stream kernel
   * starts from it (CarriedStart above; neutral_history.h: prologue_carried; identical tables, one
   * rank's whole mesh, no tile queues -- the other instantiations look up and draw as before):
   * carried_in / carried_out next to rec_in / rec_out.  `micro` is written by whoever places the
   * record (pass 0 of the stream kernel copies it across) or changes its energy (the collision
   * stage, where it ends a history), and is valid for every live slot of rec_in whenever the host
   * says so (it runs refresh_micro_kernel first when not: a store just imported, a table view
   * just rebuilt); `minus_log_rn0` of carried_in is worked out by the counting sort's placement
   * pass (pass 0) from id_in.  One 16-byte pair per slot: the stream kernel's gather is one
   * 64-byte sector per history beside the record's two, not two (the split deck, whose stream
   * kernel is nothing but refills, ran 6 % slower with two arrays: profiles/r05/experiments). */
  CarriedStart* carried_in;
  CarriedStart* carried_out;
  int carried;             /* 1: this step's stream kernel starts from them (set per step) */
  /* The graveyard.  Records [sort_end, nparticles) belong to particles that were dead when
   * the LAST step began: they keep their slots for good, in both record buffers, and take
   * no part in the sort.  The dead of this step's sort are carried over to [first_inactive,
   * sort_end) of rec_out by copy_inactive (and become graveyard next step); those carried
   * over last step, [sort_end, mirror_end), are so far in rec_in only and are copied across
   * once.  A dead particle thus costs two record copies, not one per step.  Decomposed
   * stores (whose slot population changes within a step) keep sort_end = mirror_end =
   * nparticles. */
  int sort_end;
  int mirror_end;
  int slots_by_id;         /* 1: the kernels that place a record (pass 0, copy_inactive) note
                              slot_of_id[id] -- a scattered 4-B store each, hidden in the stream
                              kernel: what the write-back goes by, in a step or on demand; 0: a
                              decomposed store, which keeps id_out[slot] instead */
  unsigned* order;         /* nparticles: record indices sorted by tile (pass 0: into rec_in,
                              the dead last; later passes: the migrants, into rec_out) */
  unsigned* tile_count;    /* nsort + 2: histogram of the counting sort (zero between uses) */
  unsigned* tile_offset;   /* nsort + 2: first sorted position of each bucket */
  unsigned* tile_cursor;   /* nsort + 2: next free position of each bucket during placement */
  uint4* chunks;           /* max_chunks: {begin, end, tile, windowed} into order[] */
  unsigned* collide_queue; /* nparticles: ids suspended at their first collision */
  /* one bit per particle id: handed to the collision stage in this step.  Set by the kernel
   * that builds the collision queue when the write-back is split (mark_suspended): the pass over
   * the ids leaves those alone -- the collision stage exports them itself -- and can therefore
   * run BESIDE the collision stage, on a stream of lowest priority. */
  unsigned* susp_ids;
  int mark_suspended;
  SuspendExtra* susp;      /* nparticles: side store of the collision stage's time slicing */
  double* susp_track;      /* nparticles, only with the scalar-flux tally (else null) */
  StealWork* steal;        /* the collision stage's rings and CU lists (neutral_kernels.hip) */
  /* The stream kernel's tile queues (neutral_tiled.hip, "asynchronous tile queue"): per tile an
   * append-only log of record slots whose histories left the tally window they were streaming
   * under with far to go and have reached this tile.  Producers reserve places with an atomic add
   * on queue_tail and store the slot (kQueueEmpty until then); workgroups claim ranges with a
   * compare-and-swap on queue_head and stream them under a window centred on the tile -- inside
   * the SAME launch: no sort, no further pass.  A tile that receives more than queue_capacity
   * in one launch overflows into the pass mechanism (kRecMigrate, sorted by the next pass).
   * null: no queues, every migrant waits for the next pass. */
  unsigned* queue_entries; /* ntiles * queue_capacity */
  unsigned* queue_tail;    /* ntiles: places reserved in this launch (may exceed the capacity) */
  unsigned* queue_head;    /* ntiles: places claimed by workgroups */
  unsigned queue_capacity;
  /* finer bucketed index for the collision stage (identical tables only; null: none) */
  const unsigned short* fine_index;
  int fine_index_n;
  long long fine_index_base;
  int fine_index_shift;
  int* edges_computed;     /* [device] 1: SolveArgs::edge_dx / edge_dy reproduce both edge arrays */
  unsigned char* tile_uniform; /* ntiles: 1 when every cell of the tile's window and of the ring
                                  around it holds one density (recomputed every step) */
  unsigned* ctrl;          /* 8 words: chunk head, #chunks, queue length, #active, #migrants,
                              passes used */
  int chunk_particles;     /* particles one workgroup takes at a time */
  int refill_min;          /* stream kernel: empty lanes of a wave that trigger a refill */
  int stream_repeat;       /* ... facet crossings of a wave per STREAM pass (scheduling only) */
  int pass;                /* 0: every live record starts its history; > 0: migrants resume */
  int allow_migrate;       /* 0 on the last permitted pass: finish with global atomics */
  double cells_per_x;      /* nx / mesh width, ny / mesh height: for the estimate of how */
  double cells_per_y;      /* many facets a history still has to cross */
  int tile_shift;          /* log2 of the tile edge in cells: 4 (16 cells) .. 7 (128) */
  int window_min_particles; /* a chunk with fewer particles tallies straight to HBM */
  int tiles_x;
  int tiles_y;
  int ntiles;
  int reach_classes;       /* 1, or 4: records are sorted by (tile, reach class) -- sparse
                              problems, neutral_history.h: reach_class */
  int nsort;               /* buckets of the counting sort: ntiles * reach_classes (+ the dead) */
  int max_chunks;
};

/* What the launches of a particle store are tuned by: constants of the kernels that tests and
 * A/B runs override from the environment, and the device's CU count.  Read ONCE per store (when
 * its records are imported), not per launch: three getenv and two runtime queries per collision
 * stage were test plumbing in the hot launch path. */
struct LaunchTuning {
  int steal_min;          /* NEUTRAL_STEAL_MIN (default kStealMin) */
  int share_weight;       /* NEUTRAL_SHARE_WEIGHT (default kOldestWeight) */
  int weighted_share_min; /* NEUTRAL_WEIGHTED_SHARE_MIN (default kWeightedShareMin) */
  int steal_delay;        /* NEUTRAL_STEAL_DELAY (default 0) */
  int max_blocks;         /* NEUTRAL_K2_MAX_BLOCKS (default 0: no cap) */
  int compute_units;
};
LaunchTuning launch_tuning_from_env();

/* spatial domain decomposition: px x py ranks, uniform blocks of bx x by cells (the
 * last ones may be smaller); rank r owns block (r % px, r / px) */
struct DomainGrid {
  int px, py;
  int bx, by;
};

/* what the host enqueues of a timestep of the tiled pipeline in one go */
struct TiledPlan {
  int stream_passes; /* stream passes to enqueue back to back (steady state: what the
                        last step needed plus one) */
  int blocks_per_cu; /* collision stage: workgroups per CU (0: as many as fit; -1: chosen by
                        the kernel from the queue length) */
};

enum Variant {
  kVariantOverParticle = 0, /* K1: one lane owns one history start to finish */
  kVariantEventSorted = 1,  /* K2: lanes re-grouped by next event */
  kVariantTiled = 2,        /* K3: tile-sorted streaming with an LDS tally window, then K2 */
};

hipError_t launch_inject(const InjectArgs& a, hipStream_t stream);
/* decomposed mesh: every rank looks at all a.nparticles candidates (positions in the
 * GLOBAL source box given by a.left_off ...) and keeps those that fall into its block
 * of the mesh (a.edgex / a.edgey are the block's edges); keys[slot] = global id,
 * *count = particles kept */
hipError_t launch_inject_filtered(const InjectArgs& a, unsigned* keys, unsigned* count,
                                  hipStream_t stream);
hipError_t launch_solve(const SolveArgs& a, int variant, hipStream_t stream);
/* The host's cached view of the two cs tables, re-checked on the device every step:
 * out[0] = 1 unless hash(scatter keys) == expect_hash_s, hash(absorb keys) ==
 * expect_hash_a and (tables element-wise identical) == expect_same; out[1], out[2] =
 * the two hashes, out[3] = the identity flag (all four words are written). */
/* out[0] |= 1 when a value lies outside [2^-100, 2^100] (zero, negative, inf, NaN too) */
hipError_t launch_unphysical_values(const double* v, long long n, unsigned long long* out,
                                    hipStream_t stream);
/* out: 16 words, the first 8 -- [0] the abort flag of the step's history kernels = [6] | [7]; [1..3]
 * hashes and identity; [4] a key or value lies outside [2^-100, 2^100]; [5] (written by
 * launch_unphysical_values before this kernel) a density does; [6] the expected hashes /
 * identity do not match; [7] fast_arithmetic was launched and [4] or [5] is set; [8..12]
 * the kernel's own accumulators (zero between launches) */
hipError_t launch_tables_check(const double* ks, const double* vs, int ns, const double* ka,
                               const double* va, int na, unsigned long long expect_hash_s,
                               unsigned long long expect_hash_a, int expect_same,
                               int fast_arithmetic, unsigned long long* out4,
                               hipStream_t stream);


/* tiled pipeline: sort by tile, stream with the LDS tally window, then K2 */
size_t tiled_lds_bytes(const SolveArgs& a, const TiledArgs& t);
/* SoA store <-> record store (ids 0..n-1 in order on import; scatter by id on export) */
hipError_t launch_import_records(const ParticleView& p, ParticleRec* rec, unsigned* info,
                                 unsigned* slot_of_id, unsigned* ids, int tiles_x, int tile_shift,
                                 int x_off, int y_off, int n, hipStream_t stream);
/* carried_in[slot].micro for every live record of t.rec_in, by plain bisection of the scatter table (no
 * index: independent of the cached view) -- after an import, after the table view was rebuilt */
hipError_t launch_refresh_micro(const SolveArgs& a, const TiledArgs& t, hipStream_t stream);
/* does this step's stream kernel start histories from the carried values? */
bool tiled_uses_carried(const SolveArgs& a, const TiledArgs& t);
/* slot_of_id: where each id's record is (TiledArgs::slot_of_id) */
/* Records in slots from a boundary on belong to particles whose final state the arrays hold
 * already and are left alone (the random access to them is what the pass is bound by):
 * first_inactive [device] -- the first slot of the particles that were dead when the step
 * began, given when the arrays were current then -- or, when that is null, final_from [host
 * value]: the graveyard's first slot at the time the arrays were last current (0xFFFFFFFF:
 * every record is written). */
hipError_t launch_export_records(const ParticleRec* rec, const unsigned* slot_of_id,
                                 const ParticleView& p, int n, hipStream_t stream,
                                 const int* abort_flag = nullptr,
                                 const unsigned* first_inactive = nullptr,
                                 unsigned final_from = 0xFFFFFFFFu,
                                 const unsigned* skip_ids = nullptr, int max_blocks = 0,
                                 bool partial_ok = false);
/* the write-back split in two (see TiledArgs::susp_ids): what launch_solve_tiled enqueues on a
 * stream of its own, beside the collision stage, when `on` */
struct SplitExport {
  bool on;
  hipStream_t side;      /* (low priority: it gets the CUs the collision stage's waves leave) */
  hipEvent_t done;       /* recorded on `side` after the pass */
  ParticleView p;
  int skip_long_dead;
};
const unsigned* tiled_first_inactive(const TiledArgs& t);
/* spatial domain decomposition (neutral_tiled.hip, section 2b): emigrants of this
 * step's records (t.rec_out) counted and packed by destination rank, arrivals appended
 * behind the first_slot records, holes closed at the end of the step (t.rec_out ->
 * t.rec_in; *cursor = records kept), and the slot-for-slot exchange with the SoA store */
hipError_t launch_emigrant_count(const TiledArgs& t, int n, const DomainGrid& d, unsigned* counts,
                                 hipStream_t stream);
hipError_t launch_emigrant_pack(const TiledArgs& t, int n, const DomainGrid& d,
                                const unsigned* offset, unsigned* cursor, ParticleRec* send,
                                unsigned* free_slots, unsigned* nfree, hipStream_t stream);
/* (the first `reuse` arrivals take the last entries of the free list of nfree slots) */
hipError_t launch_immigrant_append(const TiledArgs& t, const ParticleRec* recv, int nrecv,
                                   int first_slot, int x_off, int y_off,
                                   const unsigned* free_slots, int nfree, int reuse,
                                   hipStream_t stream);
hipError_t launch_compact_records(const TiledArgs& t, int n, unsigned* cursor, hipStream_t stream);
hipError_t launch_import_by_slot(const ParticleView& p, const unsigned* keys, const TiledArgs& t,
                                 int x_off, int y_off, int n, hipStream_t stream);
hipError_t launch_export_by_slot(const ParticleRec* rec, const ParticleView& p, unsigned* keys,
                                 int n, hipStream_t stream);
/* tile edge (log2 cells) and window threshold for a problem; tiles and chunk capacity */
int tiled_tile_shift(int nx, int ny, int nparticles, bool with_flux);
int tiled_window_min_particles(int tile_shift);
void tiled_geometry(int nx, int ny, int nparticles, int tile_shift, int* tiles_x, int* tiles_y,
                    int* max_chunks);
/* Enqueues plan.stream_passes stream passes starting with pass number first_pass, the
 * collision queue and the collision stage; nothing here waits for the device.
 * a.counters must point at TWO StepCounters records: [0] streaming kernel, [1]
 * collision kernel.  The optional events are recorded after the first sort, after
 * the last enqueued streaming pass and after the collision queue is built.  The
 * caller reads t.ctrl (migrants left over, passes used, queue length) with the
 * step's counters and calls again with first_pass = *passes_enqueued while migrants
 * are left (histories suspended by the later passes get a collision stage of their
 * own; the earlier ones are marked done).  This step's records are t.rec_out /
 * t.info_out: the caller swaps in/out when the step is complete. */
hipError_t launch_solve_tiled(const SolveArgs& a, TiledArgs& t, hipStream_t stream,
                              const TiledPlan& plan, int first_pass, hipEvent_t after_sort,
                              hipEvent_t after_stream, hipEvent_t after_collect,
                              int* passes_enqueued, const SplitExport* split = nullptr);
hipError_t launch_split_export(const SolveArgs& a, const TiledArgs& t, const SplitExport& split,
                               hipEvent_t after_collect);


/* builds start[0..nbuckets] of the bucketed index for `keys` (neutral_device.h) */
hipError_t launch_build_cs_index(const double* keys, int n, int shift, long long base,
                                 int nbuckets, unsigned short* start, hipStream_t stream);

/* unit probes of the device building blocks (all pointers [device]) */
hipError_t launch_probe_threefry(const uint64_t* in, uint64_t* out, double* rn, int n,
                                 hipStream_t stream);
struct CsIndex;
hipError_t launch_probe_cs(const double* keys, const double* values, int nentries,
                           const double* energy, double* value, int* index, int n,
                           const CsIndex& ix, hipStream_t stream);
hipError_t launch_probe_facet(const double* in, double* dist, int* x_facet, int n,
                              hipStream_t stream);
hipError_t launch_probe_division(const double* in, double* out, int* plain, int n,
                                 hipStream_t stream);
hipError_t launch_probe_log(const double* in, double* out, int n, hipStream_t stream);
hipError_t launch_probe_scatter(const double* in, double* out, int n, hipStream_t stream);

}  // namespace neutral
#endif
