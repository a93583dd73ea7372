/*
 * neutral_abi.hip -- the C ABI of libneutral_hip.so (include/neutral_hip.h):
 * the three functions of the reference's neutral_interface.h, the HBM flavour
 * of the allocation hooks, and the extension entry points.  Host code only;
 * kernels live in neutral_kernels.hip.
 */
#include "../../include/neutral_hip.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "neutral_comm.h"
#include "neutral_kernels.h"

extern "C" {
#include "../host/comms.h"
}

extern "C" {
/* host layer (neutral_amd/host/host.c), linked into this library */
int get_key_value_parameter(const char* specifier, const char* filename, char* keys,
                            double* values, int* nkeys);
int within_tolerance(const double expected, const double result, const double tolerance);
}

#define NEUTRAL_ABI_VERSION 7 /* 7: NeutralHipStepStats grew steals; 6: set_arithmetic, NeutralHipStepStats grew checked_arithmetic, attempts, host_collectives, exchange_ranks; 5: NeutralHipStepStats grew export_ms; 2: probe_division, NeutralHipStepStats grew requeued + collide_passes; 3: probe_log;
                                 4: invalidate_particles, NeutralHipStepStats grew host_syncs, stream_passes_enqueued, tile_cells */
#define NEUTRAL_MAX_KEYS 40
#define NEUTRAL_MAX_STR_LEN 1024
#define NEUTRAL_VALIDATE_TOLERANCE 1.0e-3 /* neutral_data.h:27 */

#define HIP_CHECK(expr)                                                              \
  do {                                                                               \
    hipError_t err_ = (expr);                                                        \
    if (err_ != hipSuccess) {                                                        \
      fprintf(stderr, "libneutral_hip: %s failed: %s\n%s:%d\n", #expr,               \
              hipGetErrorString(err_), __FILE__, __LINE__);                          \
      exit(EXIT_FAILURE);                                                            \
    }                                                                                \
  } while (0)

#include "neutral_device.h"

namespace {

constexpr int kMaxIndexBuckets = 16384; /* u16 entries: 32 KB of LDS at most */
/* The collision stage has the LDS to itself (three workgroups per CU): when both
 * tables are the same data it searches through an index of twice the resolution
 * (512 buckets per binade, 17 004 entries = 34 KB for the shipped table: the
 * window to bisect shrinks from 4.5 to 2.8 keys on average, from 28 to 16 at most) */
constexpr int kMaxFineIndexBuckets = 24576;

/* What the library derives from the two cs tables and keeps from step to step: are
 * they the same data (one search per energy), and the bucketed indexes over their
 * keys.  Keyed by the table pointers and sizes; the CONTENTS are re-checked on the
 * device every step (tables_check_kernel), so rewriting a table in place is noticed. */
struct TableView {
  bool valid = false;
  const double* keys_s = nullptr;
  const double* values_s = nullptr;
  int n_s = 0;
  const double* keys_a = nullptr;
  const double* values_a = nullptr;
  int n_a = 0;
  int variant = -1;
  unsigned long long hash_s = 0;
  unsigned long long hash_a = 0;
  int same = 0;
  neutral::CsIndex ix_s = {nullptr, 0, 0, 0};
  neutral::CsIndex ix_a = {nullptr, 0, 0, 0};
  neutral::CsIndex fine = {nullptr, 0, 0, 0};
};

struct StepResults {
  neutral::StepCounters counters[2];
  unsigned long long check[8];
  unsigned ctrl[16];
  unsigned long long words[16];
};

struct State {
  hipStream_t stream = nullptr;
  uint64_t pid_base = 0;
  int variant = NEUTRAL_HIP_VARIANT_TILED; /* fastest on every BASELINE deck (profiles/) */
  bool variant_from_env_done = false;
  int quiet = 0;
  char tests_file[NEUTRAL_MAX_STR_LEN] = "problems/neutral.tests"; /* neutral_data.h:33 */
  NeutralHipStepStats last = {};
  /* per-device scratch, created on first use */
  int scratch_device = -1;
  neutral::StepCounters* d_counters = nullptr;
  unsigned long long* d_check = nullptr;            /* tables_check_kernel's words
                                                       (neutral_kernels.h: launch_tables_check) */
  int arithmetic = NEUTRAL_HIP_ARITH_AUTO;          /* neutral_hip_set_arithmetic */
  bool arithmetic_from_env_done = false;
  bool use_checked = false; /* auto mode: what the last step's device-side check found ... */
  const void* checked_density = nullptr; /* ... for this density mesh (another mesh starts fast) */
  bool said_checked = false;
  neutral::ParticleView* d_export_view = nullptr;   /* the stepped store's array pointers */
  unsigned short* d_index[2] = {nullptr, nullptr}; /* bucketed cs indexes (scatter, absorb) */
  unsigned short* d_index_fine = nullptr;           /* finer index of the collision stage */
  hipEvent_t ev_start = nullptr;
  hipEvent_t ev_stop = nullptr;
  hipEvent_t ev_sorted = nullptr;   /* tiled variant: after the sort */
  hipEvent_t ev_streamed = nullptr; /* tiled variant: after the streaming kernel */
  hipEvent_t ev_collected = nullptr; /* tiled variant: after the collision queue is built */
  hipEvent_t ev_exported = nullptr; /* tiled variant: after the write-back to the SoA arrays */
  /* What the host reads at the step's single wait -- the two counter records, the check
   * words, the pipeline's control words, the step words -- lands in ONE block of pinned,
   * device-mapped host memory, written by one small kernel at the end of the batch: four
   * device-to-host copies into pageable memory cost 70-85 us each in the kernel trace
   * (r03/kernel_stats.csv: __amd_rocclr_copyBuffer), a quarter of a millisecond per step. */
  struct StepResults* h_results = nullptr; /* pinned host */
  struct StepResults* d_results = nullptr; /* the same block as the device sees it */
  hipStream_t comm_stream = nullptr; /* several ranks: the exchange runs here, beside the write-back */
  hipEvent_t ev_exchanged = nullptr;
  TableView tables;
  /* workspace of the tiled variant, grown on demand */
  neutral::TiledArgs tiled = {};
  /* which particle store the records mirror, and which copy is current */
  const void* rec_owner = nullptr; /* particles->x of the mirrored SoA store */
  neutral::ParticleView rec_owner_view = {}; /* its arrays, for the write-back */
  int rec_count = 0;
  bool rec_valid = false;          /* records hold the current state */
  bool soa_valid = true;           /* SoA arrays hold the current state */
  int lazy_export = 0;
  /* what the last step of this record store needed: the next step is enqueued on that
   * assumption, without waiting for the device in between (0 / -1: nothing known) */
  int plan_passes = 0;
  bool slots_valid = false;        /* tiled.slot_of_id describes tiled.rec_in */
  int host_syncs = 0;              /* waits for the device inside the current call */
  int host_collectives = 0;        /* collectives over the ranks' host links inside the current
                                      call, the staging of the exchange itself not counted */
  unsigned long long* d_words = nullptr; /* several ranks: the step's words (event counters,
                                            flags) that travel with the tally exchange */
  /* ranks: particle stores made by inject_particles (this rank's shards) and the
   * per-step tally that is all-reduced before it joins the caller's mesh */
  double* flux_tally = nullptr; /* scalar-flux tally of the caller (null: not kept) */
  double* d_step_flux = nullptr; /* several ranks: this step's contributions to it */
  size_t step_flux_cells = 0;
  int auto_shard = 1;
  struct Store {
    const void* key; /* particles->x */
    int count;
    uint64_t first;
    /* decomposed mesh: the store holds whatever particles are inside this rank's
     * block right now -- `count` of `capacity` slots, keys[slot] = the particle's id */
    bool decomposed;
    int capacity;
    unsigned* keys;
  };
  /* spatial domain decomposition (neutral_hip_set_decomposition) */
  bool domain_on = false;
  neutral::DomainGrid domain = {1, 1, 0, 0};
  double source_box[4] = {0.0, 0.0, 0.0, 0.0};
  bool source_box_set = false;
  unsigned* d_exchange = nullptr;    /* counts[64], offsets[64], cursors[64], 1 compaction cursor */
  neutral::ParticleRec* d_send = nullptr;
  neutral::ParticleRec* d_recv = nullptr;
  size_t send_capacity = 0; /* records */
  size_t recv_capacity = 0;
  unsigned* rec_owner_keys = nullptr; /* keys[] of the mirrored store when it is decomposed */
  unsigned* d_free_slots = nullptr;   /* slots emigrants left in this step (arrivals reuse them) */
  size_t free_slots_capacity = 0;
  int free_count = 0;
  enum { kMaxStores = 64 };
  Store stores[kMaxStores] = {};
  int nstores = 0;
  double* d_step_tally = nullptr;
  size_t step_tally_cells = 0;
  /* mesh extent: only for the tiled variant's "facets still ahead" estimate */
  double mesh_width = 1.0;
  double mesh_height = 1.0;
  const void* extent_edges = nullptr;
  double edge_dx = 0.0; /* the caller's edgedx[pad] / edgedy[pad] for the same mesh (0: none) */
  double edge_dy = 0.0;
  int extent_nx = 0;
  int extent_ny = 0;
  int tiled_particles = 0;
  int tiled_tiles = 0;
  int tiled_chunks = 0;
};

State g;

void ensure_scratch() {
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  if (g.scratch_device == dev) {
    return;
  }
  /* scratch of another device (if any) is abandoned: a process drives one GPU */
  HIP_CHECK(hipMalloc((void**)&g.d_counters, 2 * sizeof(neutral::StepCounters)));
  HIP_CHECK(hipMalloc((void**)&g.d_check, 16 * sizeof(unsigned long long)));
  HIP_CHECK(hipMemset(g.d_check, 0, 16 * sizeof(unsigned long long))); /* ([8..12]: accumulators) */
  HIP_CHECK(hipMalloc((void**)&g.d_export_view, sizeof(neutral::ParticleView)));
  HIP_CHECK(hipMalloc((void**)&g.d_exchange, sizeof(unsigned) * 200));
  HIP_CHECK(hipMalloc((void**)&g.d_words, sizeof(unsigned long long) * 16));
  g.tables.valid = false; /* its indexes live in the other device's scratch */
  HIP_CHECK(hipMalloc((void**)&g.d_index_fine,
                      sizeof(unsigned short) * (kMaxFineIndexBuckets + 1)));
  for (unsigned short*& d : g.d_index) {
    HIP_CHECK(hipMalloc((void**)&d, sizeof(unsigned short) * (kMaxIndexBuckets + 1)));
  }
  HIP_CHECK(hipEventCreate(&g.ev_start));
  HIP_CHECK(hipEventCreate(&g.ev_stop));
  HIP_CHECK(hipEventCreate(&g.ev_sorted));
  HIP_CHECK(hipEventCreate(&g.ev_streamed));
  HIP_CHECK(hipEventCreate(&g.ev_collected));
  HIP_CHECK(hipEventCreate(&g.ev_exported));
  HIP_CHECK(hipEventCreateWithFlags(&g.ev_exchanged, hipEventDisableTiming));
  HIP_CHECK(hipStreamCreateWithFlags(&g.comm_stream, hipStreamNonBlocking));
  if (!g.h_results) {
    HIP_CHECK(hipHostMalloc((void**)&g.h_results, sizeof(StepResults), hipHostMallocMapped));
    memset(g.h_results, 0, sizeof(StepResults));
  }
  HIP_CHECK(hipHostGetDevicePointer((void**)&g.d_results, g.h_results, 0));
  g.scratch_device = dev;
}

void read_variant_env() {
  if (g.variant_from_env_done) {
    return;
  }
  g.variant_from_env_done = true;
  const char* v = getenv("NEUTRAL_HIP_VARIANT");
  if (v && *v) {
    const int iv = atoi(v);
    if (iv == NEUTRAL_HIP_VARIANT_OVER_PARTICLE || iv == NEUTRAL_HIP_VARIANT_EVENT_SORTED ||
        iv == NEUTRAL_HIP_VARIANT_TILED) {
      g.variant = iv;
    } else {
      fprintf(stderr, "libneutral_hip: ignoring NEUTRAL_HIP_VARIANT=%s\n", v);
    }
  }
}

/* every wait for the device goes through here: NeutralHipStepStats.host_syncs */
void wait_for_stream() {
  HIP_CHECK(hipStreamSynchronize(g.stream));
  g.host_syncs++;
}

void* device_zalloc(size_t bytes) {
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
  HIP_CHECK(hipMemsetAsync(p, 0, bytes ? bytes : 1, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  return p;
}

/* Builds the exponent-bucketed index of one key array into d_start (see
 * neutral_device.h).  Returns a null index when the table cannot be indexed:
 * more than 65 535 entries (u16 starts) or non-positive first key (bit patterns
 * of non-positive doubles do not order like their values). */
neutral::CsIndex build_index(const double* d_keys, int n, unsigned short* d_start,
                             int first_shift = 44 /* 256 buckets per binade */,
                             int max_buckets = kMaxIndexBuckets) {
  neutral::CsIndex ix = {nullptr, 0, 0, 0};
  if (n < 2 || n > 65535) {
    return ix;
  }
  double ends[2];
  HIP_CHECK(hipMemcpyAsync(&ends[0], d_keys, sizeof(double), hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipMemcpyAsync(&ends[1], d_keys + (n - 1), sizeof(double), hipMemcpyDeviceToHost,
                           g.stream));
  wait_for_stream();
  if (!(ends[0] > 0.0) || !(ends[1] > ends[0])) {
    return ix;
  }
  long long lo_bits, hi_bits;
  memcpy(&lo_bits, &ends[0], sizeof(lo_bits));
  memcpy(&hi_bits, &ends[1], sizeof(hi_bits));
  int shift = first_shift;
  while (((hi_bits >> shift) - (lo_bits >> shift) + 1) > max_buckets) {
    shift++;
  }
  ix.shift = shift;
  ix.base = lo_bits >> shift;
  ix.nbuckets = (int)((hi_bits >> shift) - ix.base + 1);
  HIP_CHECK(neutral::launch_build_cs_index(d_keys, n, ix.shift, ix.base, ix.nbuckets, d_start,
                                           g.stream));
  ix.start = d_start;
  return ix;
}

/* Writes the records back to the SoA store they mirror if they are ahead of it.
 * Safe to call with any (or no) store in hand: the owner's arrays are remembered. */
void sync_soa() {
  if (!g.soa_valid && g.rec_valid && g.rec_owner) {
    if (g.rec_owner_keys) { /* decomposed mesh: slot for slot */
      HIP_CHECK(neutral::launch_export_by_slot(g.tiled.rec_in, g.rec_owner_view, g.rec_owner_keys,
                                               g.rec_count, g.stream));
    } else {
      if (!g.slots_valid) { /* (the steps since did not keep slot_of_id) */
        HIP_CHECK(neutral::launch_invert_ids(g.tiled.rec_in, g.tiled.slot_of_id, g.rec_count,
                                             g.stream));
        g.slots_valid = true;
      }
      HIP_CHECK(neutral::launch_export_records(g.tiled.rec_in, g.tiled.slot_of_id,
                                               g.rec_owner_view, g.rec_count, g.stream));
    }
    wait_for_stream();
  }
  g.soa_valid = true;
}

/* The records no longer mirror their SoA store (it was rewritten, or the record
 * layout changes): the next tiled step imports it again. */
void drop_records() {
  g.rec_valid = false;
  g.free_count = 0; /* (slots emigrants left are holes of the records, not of the arrays) */
  g.plan_passes = 0;
}

/* (Re)allocates the tiled variant's workspace for this problem size. */
void ensure_tiled_workspace(int nx, int ny, int nparticles_now, int capacity) {
  /* (a decomposed store can grow up to its capacity within a step: buffers are sized
   * for that, the tile edge for what is there now) */
  int tx, ty, max_chunks;
  const int shift = neutral::tiled_tile_shift(nx, ny, nparticles_now, g.flux_tally != nullptr);
  const int nparticles = capacity > nparticles_now ? capacity : nparticles_now;
  neutral::tiled_geometry(nx, ny, nparticles, shift, &tx, &ty, &max_chunks);
  neutral::TiledArgs& t = g.tiled;
  const bool grow = nparticles > g.tiled_particles || tx * ty > g.tiled_tiles;
  if (grow || shift != t.tile_shift || tx != t.tiles_x || ty != t.tiles_y) {
    /* the record summaries hold tile numbers of the old geometry, and the buffers
     * may be about to go: a pending write-back of their owner comes first */
    sync_soa();
    drop_records();
  }
  if (g.flux_tally && (grow || !t.susp_track)) {
    /* pending weight * path length of time-sliced histories (scalar flux only) */
    if (t.susp_track) HIP_CHECK(hipFree(t.susp_track));
    const size_t cap = (size_t)(grow ? nparticles : g.tiled_particles);
    HIP_CHECK(hipMalloc((void**)&t.susp_track, sizeof(double) * (cap ? cap : 1)));
  }
  if (grow) {
    void* old[] = {t.order,  t.collide_queue, t.tile_count, t.tile_offset, t.tile_cursor, t.rec_in,
                   t.rec_out, t.info_in,      t.info_out,   t.susp,        t.id_in,       t.id_out,
                   t.slot_of_id, t.tile_uniform};
    for (void* p : old) {
      if (p) HIP_CHECK(hipFree(p));
    }
    const size_t n = (size_t)nparticles;
    const size_t nb = (size_t)(tx * ty * 4 + 2); /* (up to four reach classes per tile) */
    HIP_CHECK(hipMalloc((void**)&t.order, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.collide_queue, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.rec_in, sizeof(neutral::ParticleRec) * n));
    HIP_CHECK(hipMalloc((void**)&t.rec_out, sizeof(neutral::ParticleRec) * n));
    HIP_CHECK(hipMalloc((void**)&t.info_in, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.info_out, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.id_in, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.id_out, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.slot_of_id, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.susp, sizeof(neutral::SuspendExtra) * n));
    HIP_CHECK(hipMalloc((void**)&t.tile_count, sizeof(unsigned) * nb));
    HIP_CHECK(hipMalloc((void**)&t.tile_offset, sizeof(unsigned) * nb));
    HIP_CHECK(hipMalloc((void**)&t.tile_cursor, sizeof(unsigned) * nb));
    HIP_CHECK(hipMalloc((void**)&t.tile_uniform, (size_t)(tx * ty + 1)));
    HIP_CHECK(hipMemsetAsync(t.tile_uniform, 0, (size_t)(tx * ty + 1), g.stream));
    g.tiled_particles = nparticles;
    g.tiled_tiles = tx * ty;
  }
  /* the counting sort expects its histogram zeroed (it clears what it consumes) */
  HIP_CHECK(hipMemsetAsync(t.tile_count, 0, sizeof(unsigned) * (size_t)(tx * ty * 4 + 2), g.stream));
  if (max_chunks > g.tiled_chunks) {
    if (t.chunks) HIP_CHECK(hipFree(t.chunks));
    HIP_CHECK(hipMalloc((void**)&t.chunks, sizeof(uint4) * (size_t)max_chunks));
    g.tiled_chunks = max_chunks;
  }
  if (!t.ctrl) {
    HIP_CHECK(hipMalloc((void**)&t.ctrl, sizeof(unsigned) * 16));
    HIP_CHECK(hipMemsetAsync(t.ctrl, 0, sizeof(unsigned) * 16, g.stream));
    HIP_CHECK(hipMalloc((void**)&t.edges_computed, sizeof(int)));
    HIP_CHECK(hipMemsetAsync(t.edges_computed, 0, sizeof(int), g.stream));
  }
  t.tile_shift = shift;
  t.window_min_particles = neutral::tiled_window_min_particles(shift);
  t.tiles_x = tx;
  t.tiles_y = ty;
  t.ntiles = tx * ty;
  /* Sparse problems (about a workgroup's worth of particles per tile and pass, or fewer)
   * also sort by reach class inside a tile (neutral_history.h: reach_class); where tiles hold
   * tens of thousands, lanes are refilled from the chunk and the order inside it does not
   * matter.  NEUTRAL_REACH_CLASSES=1|4 overrides. */
  {
    const long long per_tile = (long long)nparticles_now / (tx * ty > 0 ? tx * ty : 1);
    /* (and only while the buckets still fit the sort's LDS histogram: 8 192) */
    int classes = (per_tile < 8192 && (long long)tx * ty * 4 + 1 <= 8192) ? 4 : 1;
    const char* force = getenv("NEUTRAL_REACH_CLASSES");
    if (force && (atoi(force) == 1 || atoi(force) == 4)) {
      classes = atoi(force);
    }
    if (classes != t.reach_classes && t.reach_classes != 0) {
      sync_soa();
      drop_records(); /* (the summaries' class field changes meaning) */
    }
    t.reach_classes = classes;
    t.nsort = t.ntiles * classes;
  }
  t.max_chunks = max_chunks;
}

/* true when [p, p + bytes) overlaps one of the arrays of the store the records mirror */
bool touches_record_owner(const void* p, size_t bytes) {
  if (!g.rec_owner || !g.rec_valid) {
    return false;
  }
  const neutral::ParticleView& v = g.rec_owner_view;
  const size_t n = (size_t)g.rec_count;
  const char* lo = (const char*)p;
  const char* hi = lo + bytes;
  const void* f64[] = {v.x, v.y, v.omega_x, v.omega_y, v.energy, v.weight, v.dt_to_census,
                       v.mfp_to_collision};
  for (const void* a : f64) {
    if (lo < (const char*)a + sizeof(double) * n && hi > (const char*)a) return true;
  }
  const void* i32[] = {v.cellx, v.celly, v.dead};
  for (const void* a : i32) {
    if (lo < (const char*)a + sizeof(int) * n && hi > (const char*)a) return true;
  }
  return false;
}

/* a caller is about to overwrite device memory through one of the library's own
 * copy hooks: if it is part of the mirrored particle store, the store becomes the
 * truth again (pending record state is written back first, so a partial overwrite
 * keeps the rest) */
void before_device_write(const void* dst, size_t bytes) {
  if (touches_record_owner(dst, bytes)) {
    sync_soa();
    drop_records();
  }
}

/* What the library derives from the cs tables (see TableView).  Builds the view when
 * the tables (pointers, sizes, variant) are new -- that waits for the device -- and
 * otherwise only enqueues the device-side check of the contents. */
void refresh_table_view(const NeutralHipCrossSection* cs_s, const NeutralHipCrossSection* cs_a,
                        bool rebuild, bool fast_arithmetic) {
  TableView& v = g.tables;
  const bool same_args = v.valid && v.keys_s == cs_s->keys && v.values_s == cs_s->values &&
                         v.n_s == cs_s->nentries && v.keys_a == cs_a->keys &&
                         v.values_a == cs_a->values && v.n_a == cs_a->nentries &&
                         v.variant == g.variant;
  if (!same_args || rebuild) {
    v.valid = false;
    v.keys_s = cs_s->keys;
    v.values_s = cs_s->values;
    v.n_s = cs_s->nentries;
    v.keys_a = cs_a->keys;
    v.values_a = cs_a->values;
    v.n_a = cs_a->nentries;
    v.variant = g.variant;
    /* identity and key hashes from the check kernel itself (expectations unknown) */
    HIP_CHECK(neutral::launch_tables_check(v.keys_s, v.values_s, v.n_s, v.keys_a, v.values_a, v.n_a,
                                           0ull, 0ull, -1, 0, g.d_check, g.stream));
    unsigned long long h[4];
    HIP_CHECK(hipMemcpyAsync(h, g.d_check, sizeof(h), hipMemcpyDeviceToHost, g.stream));
    wait_for_stream();
    v.hash_s = h[1];
    v.hash_a = h[2];
    v.same = (int)h[3];
    /* bucketed indexes */
    v.ix_s = build_index(v.keys_s, v.n_s, g.d_index[0]);
    v.ix_a = v.ix_s;
    if (!v.same) {
      v.ix_a = build_index(v.keys_a, v.n_a, g.d_index[1]);
      if (v.ix_a.start && v.ix_s.start && v.ix_a.shift != v.ix_s.shift) {
        v.ix_a.start = nullptr; /* one shift per launch: the absorb table falls back to bisection */
      }
      if (!v.ix_s.start && v.ix_a.start) {
        v.ix_s.shift = v.ix_a.shift;
      }
    }
    v.fine = {nullptr, 0, 0, 0};
    if (v.same && v.ix_s.start && g.variant == NEUTRAL_HIP_VARIANT_TILED) {
      const neutral::CsIndex fine =
          build_index(v.keys_s, v.n_s, g.d_index_fine, 43, kMaxFineIndexBuckets);
      if (fine.start && fine.shift < v.ix_s.shift) {
        v.fine = fine;
      }
    }
    v.valid = true;
  }
  /* every step: the contents against the view (result read with the step's counters) */
  HIP_CHECK(neutral::launch_tables_check(v.keys_s, v.values_s, v.n_s, v.keys_a, v.values_a, v.n_a,
                                         v.hash_s, v.hash_a, v.same, fast_arithmetic ? 1 : 0,
                                         g.d_check, g.stream));
}

neutral::ParticleView view_of(const NeutralHipParticle* p) {
  neutral::ParticleView v;
  v.x = p->x;
  v.y = p->y;
  v.omega_x = p->omega_x;
  v.omega_y = p->omega_y;
  v.energy = p->energy;
  v.weight = p->weight;
  v.dt_to_census = p->dt_to_census;
  v.mfp_to_collision = p->mfp_to_collision;
  v.cellx = p->cellx;
  v.celly = p->celly;
  v.dead = p->dead;
  return v;
}

/* What the ranks need of each other per batch of launches besides the tally: the event
 * counters and the flags every rank must act on together (an attempt turned down, stream
 * passes still owed).  Packed on the device, summed by the same transport as the tally on
 * the same stream, read with the batch's single wait: a steady-state step makes no
 * collective over the host links of its own. */
enum StepWord : int {
  kWordCounters = 0,   /* 2 x {nprocessed, nfacets, ncollisions, ncensus} */
  kWordRequeued = 8,
  kWordCollidePasses = 9,
  kWordTurnedDown = 10, /* ranks whose attempt was turned down on the device */
  kWordMigrants = 11,   /* histories still waiting for a stream pass */
  kWordQueued = 12,     /* histories this batch's collision stage was handed */
  kWordAborted = 13,
  kWordRanks = 14,      /* 1 per rank: how many ranks the transport summed over */
  kWordSteals = 15,     /* rings the collision stage's waves took from (see StepCounters) */
  kStepWords = 16,
};

/* the step's results, into the pinned block the host reads after its wait (one workgroup) */
__global__ void publish_results_kernel(const neutral::StepCounters* counters,
                                       const unsigned long long* check, const unsigned* ctrl,
                                       const unsigned long long* words, StepResults* out) {
  const unsigned t = threadIdx.x;
  const unsigned* c32 = (const unsigned*)counters;
  unsigned* o32 = (unsigned*)out->counters;
  for (unsigned i = t; i < 2 * sizeof(neutral::StepCounters) / 4; i += blockDim.x) {
    o32[i] = c32[i];
  }
  if (t < 8) out->check[t] = check[t];
  if (t < 16) out->ctrl[t] = ctrl ? ctrl[t] : 0u;
  if (t < 16) out->words[t] = words ? words[t] : 0ull;
}

__global__ void pack_step_words_kernel(const neutral::StepCounters* c, const unsigned long long* check,
                                       const unsigned* ctrl, unsigned long long* w) {
  if (threadIdx.x != 0) {
    return;
  }
  for (int k = 0; k < 2; ++k) {
    w[kWordCounters + 4 * k + 0] = c[k].nprocessed;
    w[kWordCounters + 4 * k + 1] = c[k].nfacets;
    w[kWordCounters + 4 * k + 2] = c[k].ncollisions;
    w[kWordCounters + 4 * k + 3] = c[k].ncensus;
  }
  w[kWordRequeued] = c[0].nrequeued + c[1].nrequeued;
  w[kWordCollidePasses] = c[0].ncollide_passes + c[1].ncollide_passes;
  w[kWordTurnedDown] = check[0] ? 1ull : 0ull;
  w[kWordMigrants] = ctrl ? ctrl[4] : 0u;
  w[kWordQueued] = ctrl ? ctrl[2] : 0u;
  w[kWordAborted] = (unsigned long long)c[0].aborted + c[1].aborted;
  w[kWordRanks] = 1ull;
  w[kWordSteals] = c[0].nsteals + c[1].nsteals;
}

__global__ void add_step_tally_kernel(double* __restrict__ tally, const double* __restrict__ step,
                                      size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    tally[i] += step[i];
  }
}

/* End of a timestep with several ranks: ONE all-reduce of the step's tally
 * contributions (sum, f64, nx*ny) on the kernels' stream; the sum joins the caller's
 * mesh, which then holds the same global tally on every rank.  The step buffer is
 * cleared again, so a step that needs more stream passes than were enqueued simply
 * exchanges what those add. */
void exchange_step(const neutral::SolveArgs& a, double* tally, bool tiled) {
  /* on a stream of its own, after the step's kernels (g.ev_stop) and BESIDE the write-back
   * of the records that the caller enqueues next on its own stream; finish_exchange() joins */
  hipStream_t xs = g.comm_stream;
  HIP_CHECK(hipStreamWaitEvent(xs, g.ev_stop, 0));
  const size_t ncells = (size_t)a.nx * (size_t)a.ny;
  hipLaunchKernelGGL(pack_step_words_kernel, dim3(1), dim3(64), 0, xs, g.d_counters, g.d_check,
                     tiled ? (const unsigned*)g.tiled.ctrl : (const unsigned*)nullptr, g.d_words);
  HIP_CHECK(hipGetLastError());
  neutral::comm_allreduce_sum(g.d_words, (size_t)kStepWords, false, xs);
  neutral::comm_allreduce_sum(a.tally, ncells, true, xs);
  hipLaunchKernelGGL(add_step_tally_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0,
                     xs, tally, (const double*)a.tally, ncells);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemsetAsync(a.tally, 0, sizeof(double) * ncells, xs));
  if (g.flux_tally) { /* the scalar-flux mesh travels the same way */
    neutral::comm_allreduce_sum(a.flux_tally, ncells, true, xs);
    hipLaunchKernelGGL(add_step_tally_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0,
                       xs, g.flux_tally, (const double*)a.flux_tally, ncells);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemsetAsync(a.flux_tally, 0, sizeof(double) * ncells, xs));
  }
  HIP_CHECK(hipEventRecord(g.ev_exchanged, xs));
}

/* the caller's stream goes on only when the exchange is done (the step buffers are reused) */
void finish_exchange() { HIP_CHECK(hipStreamWaitEvent(g.stream, g.ev_exchanged, 0)); }

/* enqueues the publication of the batch's results; fetch_results() after the wait */
void publish_results(bool tiled, bool with_words) {
  hipLaunchKernelGGL(publish_results_kernel, dim3(1), dim3(64), 0, g.stream, g.d_counters, g.d_check,
                     tiled ? (const unsigned*)g.tiled.ctrl : (const unsigned*)nullptr,
                     with_words ? (const unsigned long long*)g.d_words
                                : (const unsigned long long*)nullptr,
                     g.d_results);
  HIP_CHECK(hipGetLastError());
}

void fetch_results(neutral::StepCounters* hc, unsigned long long* check, unsigned* ctrl,
                   unsigned long long* words) {
  const StepResults& r = *g.h_results;
  memcpy(hc, r.counters, sizeof(r.counters));
  if (check) memcpy(check, r.check, sizeof(r.check));
  if (ctrl) memcpy(ctrl, r.ctrl, sizeof(r.ctrl));
  if (words) memcpy(words, r.words, sizeof(r.words));
}

/* Decomposed mesh, one round: this rank's emigrants (records of t.rec_out marked
 * kRecEmigrate) go to the ranks that own the cells they crossed into; what arrives
 * is appended behind the a.nparticles records already here, as migrants.  Returns the
 * number of arrivals.  Collective over the ranks. */
int exchange_particles(const neutral::SolveArgs& a, neutral::TiledArgs& t) {
  const int n = neutral::comm_nranks();
  const int me = neutral::comm_rank();
  if (n > 64) {
    fprintf(stderr, "libneutral_hip: the decomposed-mesh exchange handles up to 64 ranks.\n");
    exit(EXIT_FAILURE);
  }
  unsigned* d_counts = g.d_exchange;
  unsigned* d_offsets = g.d_exchange + 64;
  unsigned* d_cursor = g.d_exchange + 128;
  HIP_CHECK(hipMemsetAsync(g.d_exchange, 0, sizeof(unsigned) * 192, g.stream));
  HIP_CHECK(neutral::launch_emigrant_count(t, a.nparticles, g.domain, d_counts, g.stream));
  unsigned counts[64];
  HIP_CHECK(hipMemcpyAsync(counts, d_counts, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost,
                           g.stream));
  wait_for_stream();
  uint64_t matrix[64 * 64];
  memset(matrix, 0, sizeof(uint64_t) * (size_t)n * n);
  unsigned offsets[64];
  size_t out = 0;
  for (int d = 0; d < n; ++d) {
    offsets[d] = (unsigned)out;
    out += counts[d];
    matrix[(size_t)me * n + d] = (uint64_t)counts[d] * sizeof(neutral::ParticleRec);
  }
  if (counts[me] != 0) {
    fprintf(stderr, "libneutral_hip: rank %d: %u emigrants are bound for their own rank (the "
                    "decomposition given to neutral_hip_set_decomposition does not match the "
                    "mesh blocks passed to solve_transport_2d).\n", me, counts[me]);
    exit(EXIT_FAILURE);
  }
  comms_allreduce_u64(matrix, (size_t)n * n, COMMS_SUM);
  g.host_collectives++;
  size_t in = 0;
  for (int s2 = 0; s2 < n; ++s2) {
    in += (size_t)(matrix[(size_t)s2 * n + me] / sizeof(neutral::ParticleRec));
  }
  if (out > g.send_capacity) {
    if (g.d_send) HIP_CHECK(hipFree(g.d_send));
    g.send_capacity = out + out / 2 + 1024;
    HIP_CHECK(hipMalloc((void**)&g.d_send, sizeof(neutral::ParticleRec) * g.send_capacity));
  }
  if (in > g.recv_capacity) {
    if (g.d_recv) HIP_CHECK(hipFree(g.d_recv));
    g.recv_capacity = in + in / 2 + 1024;
    HIP_CHECK(hipMalloc((void**)&g.d_recv, sizeof(neutral::ParticleRec) * g.recv_capacity));
  }
  if ((size_t)g.tiled_particles > g.free_slots_capacity) {
    if (g.d_free_slots) HIP_CHECK(hipFree(g.d_free_slots));
    g.free_slots_capacity = (size_t)g.tiled_particles;
    HIP_CHECK(hipMalloc((void**)&g.d_free_slots, sizeof(unsigned) * g.free_slots_capacity));
  }
  /* the free list's length lives on the device next to the other exchange words */
  unsigned* d_nfree = g.d_exchange + 193;
  const unsigned nfree_now = (unsigned)g.free_count;
  HIP_CHECK(hipMemcpyAsync(d_nfree, &nfree_now, sizeof(unsigned), hipMemcpyHostToDevice, g.stream));
  HIP_CHECK(hipMemcpyAsync(d_offsets, offsets, sizeof(unsigned) * (size_t)n, hipMemcpyHostToDevice,
                           g.stream));
  HIP_CHECK(neutral::launch_emigrant_pack(t, a.nparticles, g.domain, d_offsets, d_cursor, g.d_send,
                                          g.d_free_slots, d_nfree, g.stream));
  g.free_count += (int)out;
  neutral::comm_exchange_bytes(g.d_send, g.d_recv, matrix, g.stream);
  g.host_syncs++;
  /* arrivals take the slots emigrants left first, then slots behind the records */
  const int reuse = ((int)in < g.free_count) ? (int)in : g.free_count;
  const int grow = (int)in - reuse;
  if ((size_t)a.nparticles + (size_t)grow > (size_t)g.tiled_particles) {
    fprintf(stderr, "libneutral_hip: rank %d: %zu particles arrive but the store is full (%d "
                    "slots).\n", me, in, g.tiled_particles);
    exit(EXIT_FAILURE);
  }
  HIP_CHECK(neutral::launch_immigrant_append(t, g.d_recv, (int)in, a.nparticles, a.x_off, a.y_off,
                                             g.d_free_slots, g.free_count, reuse, g.stream));
  g.free_count -= reuse;
  return grow;
}

const State::Store* find_store(const NeutralHipParticle* p) {
  for (int i = 0; i < g.nstores; ++i) {
    if (p && g.stores[i].key == (const void*)p->x) {
      return &g.stores[i];
    }
  }
  return nullptr;
}

State::Store* remember_store(const NeutralHipParticle* p, int count, uint64_t first) {
  if (g.nstores == State::kMaxStores) {
    fprintf(stderr, "libneutral_hip: more than %d sharded particle stores alive at once "
                    "(neutral_hip_free_particles releases one).\n", (int)State::kMaxStores);
    exit(EXIT_FAILURE);
  }
  const int slot = g.nstores++;
  g.stores[slot] = State::Store{(const void*)p->x, count, first, false, count, nullptr};
  return &g.stores[slot];
}

void forget_store(const NeutralHipParticle* p) {
  for (int i = 0; i < g.nstores; ++i) {
    if (g.stores[i].key == (const void*)p->x) {
      if (g.stores[i].keys) HIP_CHECK(hipFree(g.stores[i].keys));
      g.stores[i] = g.stores[--g.nstores];
      return;
    }
  }
}

double* step_flux(size_t ncells) {
  if (ncells > g.step_flux_cells) {
    if (g.d_step_flux) HIP_CHECK(hipFree(g.d_step_flux));
    HIP_CHECK(hipMalloc((void**)&g.d_step_flux, sizeof(double) * ncells));
    g.step_flux_cells = ncells;
  }
  HIP_CHECK(hipMemsetAsync(g.d_step_flux, 0, sizeof(double) * ncells, g.stream));
  return g.d_step_flux;
}

/* this step's tally contributions when several ranks share the problem */
double* step_tally(size_t ncells) {
  if (ncells > g.step_tally_cells) {
    if (g.d_step_tally) HIP_CHECK(hipFree(g.d_step_tally));
    HIP_CHECK(hipMalloc((void**)&g.d_step_tally, sizeof(double) * ncells));
    g.step_tally_cells = ncells;
  }
  HIP_CHECK(hipMemsetAsync(g.d_step_tally, 0, sizeof(double) * ncells, g.stream));
  return g.d_step_tally;
}

void run_inject(const int nparticles, const int local_nx, const int local_ny, const int pad,
                const double left_off, const double bottom_off, const double width,
                const double height, const int x_off, const int y_off, const double dt,
                const double* edgex, const double* edgey, const double initial_energy,
                const NeutralHipParticle* particles) {
  neutral::InjectArgs a;
  a.nparticles = nparticles;
  a.pid_base = g.pid_base;
  a.local_nx = local_nx;
  a.local_ny = local_ny;
  a.pad = pad;
  a.x_off = x_off;
  a.y_off = y_off;
  a.left_off = left_off;
  a.bottom_off = bottom_off;
  a.width = width;
  a.height = height;
  a.dt = dt;
  a.initial_energy = initial_energy;
  a.edgex = edgex;
  a.edgey = edgey;
  a.p = view_of(particles);
  if (g.rec_owner == (const void*)particles->x) {
    drop_records(); /* the SoA store is about to be rewritten */
    g.soa_valid = true;
  }
  HIP_CHECK(neutral::launch_inject(a, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}

/* decomposed mesh: this rank's part of the injected particles (see inject_particles) */
void run_inject_filtered(State::Store* st, const int nparticles, const int local_nx,
                         const int local_ny, const int pad, const double left_off,
                         const double bottom_off, const double width, const double height,
                         const int x_off, const int y_off, const double dt, const double* edgex,
                         const double* edgey, const double initial_energy,
                         const NeutralHipParticle* particles) {
  ensure_scratch();
  neutral::InjectArgs a;
  a.nparticles = nparticles;
  a.pid_base = 0;
  a.local_nx = local_nx;
  a.local_ny = local_ny;
  a.pad = pad;
  a.x_off = x_off;
  a.y_off = y_off;
  /* the global source box, if the caller named it (neutral_hip_set_source_box);
   * otherwise the box passed in is taken to be it */
  a.left_off = g.source_box_set ? g.source_box[0] : left_off;
  a.bottom_off = g.source_box_set ? g.source_box[1] : bottom_off;
  a.width = g.source_box_set ? g.source_box[2] : width;
  a.height = g.source_box_set ? g.source_box[3] : height;
  a.dt = dt;
  a.initial_energy = initial_energy;
  a.edgex = edgex;
  a.edgey = edgey;
  a.p = view_of(particles);
  if (g.rec_owner == (const void*)particles->x) {
    drop_records();
    g.soa_valid = true;
  }
  unsigned kept = 0;
  HIP_CHECK(neutral::launch_inject_filtered(a, st->keys, g.d_exchange + 192, g.stream));
  HIP_CHECK(hipMemcpyAsync(&kept, g.d_exchange + 192, sizeof(unsigned), hipMemcpyDeviceToHost,
                           g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  st->count = (int)kept;
}

}  // namespace

extern "C" {

/* ---- 1. reference kernel interface ------------------------------------------ */

void solve_transport_2d(const int nx, const int ny, const int global_nx, const int global_ny,
                        const uint64_t master_key, const int pad, const int x_off,
                        const int y_off, const double dt, const int ntotal_particles,
                        int* nlocal_particles, const int* neighbours,
                        NeutralHipParticle* particles, const double* density,
                        const double* edgex, const double* edgey, const double* edgedx,
                        const double* edgedy, NeutralHipCrossSection* cs_scatter_table,
                        NeutralHipCrossSection* cs_absorb_table,
                        double* energy_deposition_tally, uint64_t* reduce_array0,
                        uint64_t* reduce_array1, uint64_t* reduce_array2,
                        uint64_t* facet_events, uint64_t* collision_events) {
  (void)neighbours;
  (void)reduce_array0;
  (void)reduce_array1;
  (void)reduce_array2;

  if (!(*nlocal_particles) && neutral::comm_nranks() == 1) {
    printf("Out of particles\n"); /* omp3/neutral.c:30-33 */
    fflush(stdout);
    return;
  }
  /* (with several ranks a rank without particles still takes part: the exchanges at
   * the end of the step are collective, and on a decomposed mesh particles may arrive) */
  if (!particles || !particles->x || !particles->dead) {
    fprintf(stderr, "libneutral_hip: solve_transport_2d needs a particle store made by "
                    "this library's inject_particles (the dead[] array is required).\n");
    exit(EXIT_FAILURE);
  }
  if (cs_scatter_table->nentries < 2 || cs_absorb_table->nentries < 2) {
    fprintf(stderr, "libneutral_hip: cross-section tables need at least 2 entries.\n");
    exit(EXIT_FAILURE);
  }

  read_variant_env();
  if (!g.arithmetic_from_env_done) {
    g.arithmetic_from_env_done = true;
    const char* arith = getenv("NEUTRAL_HIP_ARITH");
    if (arith && strcmp(arith, "checked") == 0) {
      g.arithmetic = NEUTRAL_HIP_ARITH_CHECKED;
    }
  }
  ensure_scratch();
  g.host_syncs = 0;
  const bool tiled = (g.variant == NEUTRAL_HIP_VARIANT_TILED);
  if (tiled && pad != 0) {
    fprintf(stderr, "libneutral_hip: the tiled variant needs pad = 0 (as main.c:33 sets).\n");
    exit(EXIT_FAILURE);
  }

  neutral::SolveArgs a;
  a.nx = nx;
  a.ny = ny;
  a.global_nx = global_nx;
  a.global_ny = global_ny;
  a.master_key = master_key;
  a.pad = pad;
  a.x_off = x_off;
  a.y_off = y_off;
  a.dt = dt;
  a.inv_ntotal_particles = 1.0 / (double)ntotal_particles; /* omp3/neutral.c:120 */
  /* several ranks: the store holds this rank's shard (inject_particles made it so), or
   * -- decomposed mesh -- the particles that are inside this rank's block right now */
  State::Store* shard = const_cast<State::Store*>(find_store(particles));
  const bool decomposed = shard && shard->decomposed;
  if (!decomposed && neutral::comm_nranks() == 1) {
    shard = nullptr;
  }
  if (decomposed && !tiled) {
    fprintf(stderr, "libneutral_hip: a decomposed mesh needs the tiled variant.\n");
    exit(EXIT_FAILURE);
  }
  a.nparticles = shard ? shard->count : *nlocal_particles;
  a.pid_base = decomposed ? 0 : (shard ? shard->first : g.pid_base);
  a.p = view_of(particles);
  a.density = density;
  a.edgex = edgex;
  a.edgey = edgey;
  a.edge_dx = 0.0;
  a.edge_dy = 0.0;
  a.tally = energy_deposition_tally;
  a.flux_tally = g.flux_tally;
  a.susp_track = nullptr;
  a.counters = g.d_counters;
  a.queue = nullptr;
  a.queue_len = nullptr;
  a.rec = nullptr;
  a.blocks_per_cu = 0;
  a.max_blocks = 0;
  a.slot_info = nullptr;
  a.tiles_x = 0;
  a.tile_shift = 4;
  a.susp = nullptr;
  /* Default (eager) mode: the SoA arrays are current when the call returns.  One
   * export pass at the end of the step does that (11 ms at 1e8 particles); letting
   * every kernel that ends a history store it to the arrays itself -- eleven
   * scattered 8-byte stores per history -- costs 18 ms (profiles/r02: fused export)
   * and stays an experiment: NEUTRAL_HIP_FUSED_EXPORT=1. */
  const char* fused_env = getenv("NEUTRAL_HIP_FUSED_EXPORT");
  const bool fused_export = tiled && !g.lazy_export && fused_env && atoi(fused_env) != 0;
  const bool pass_export = tiled && !g.lazy_export && (!fused_export || decomposed);
  a.export_soa = (fused_export && !decomposed) ? 1 : 0;
  a.export_view = nullptr;
  a.export_skip_long_dead = 0;
  a.decomposed = decomposed ? 1 : 0;
  a.emigrants = nullptr;
  a.abort_flag = (const int*)g.d_check; /* low word of tables_check_kernel's verdict */

  if (tiled) {
    ensure_tiled_workspace(nx, ny, a.nparticles, decomposed ? shard->capacity : 0);
    /* the records mirror one SoA store: (re)import when they are not current */
    if (!g.rec_valid || g.rec_owner != (const void*)particles->x ||
        g.rec_count != a.nparticles) {
      sync_soa(); /* a previous owner's pending write-back */
      drop_records();
      if (decomposed) {
        HIP_CHECK(neutral::launch_import_by_slot(a.p, shard->keys, g.tiled, x_off, y_off,
                                                 a.nparticles, g.stream));
      } else {
        HIP_CHECK(neutral::launch_import_records(a.p, g.tiled.rec_in, g.tiled.info_in,
                                                 g.tiled.slot_of_id, g.tiled.tiles_x,
                                                 g.tiled.tile_shift, x_off, y_off, a.nparticles,
                                                 g.stream));
      }
      g.slots_valid = !decomposed; /* (the import lays the records out by id) */
      g.tiled.sort_end = a.nparticles; /* (no graveyard yet) */
      g.tiled.mirror_end = a.nparticles;
      g.rec_owner = (const void*)particles->x;
      g.rec_owner_view = a.p;
      g.rec_owner_keys = decomposed ? shard->keys : nullptr;
      g.rec_count = a.nparticles;
      g.rec_valid = true;
    }
    if (g.extent_edges != (const void*)edgex || g.extent_nx != nx || g.extent_ny != ny) {
      /* mesh extent from the edge arrays (four doubles, once per mesh) */
      double e[4];
      HIP_CHECK(hipMemcpyAsync(&e[0], edgex + pad, sizeof(double), hipMemcpyDeviceToHost, g.stream));
      HIP_CHECK(hipMemcpyAsync(&e[1], edgex + pad + nx, sizeof(double), hipMemcpyDeviceToHost,
                               g.stream));
      HIP_CHECK(hipMemcpyAsync(&e[2], edgey + pad, sizeof(double), hipMemcpyDeviceToHost, g.stream));
      HIP_CHECK(hipMemcpyAsync(&e[3], edgey + pad + ny, sizeof(double), hipMemcpyDeviceToHost,
                               g.stream));
      /* (and the spacings the host layer made the edges from, if the caller passes them) */
      double d[2] = {0.0, 0.0};
      if (edgedx && edgedy) {
        HIP_CHECK(hipMemcpyAsync(&d[0], edgedx + pad, sizeof(double), hipMemcpyDeviceToHost, g.stream));
        HIP_CHECK(hipMemcpyAsync(&d[1], edgedy + pad, sizeof(double), hipMemcpyDeviceToHost, g.stream));
      }
      wait_for_stream();
      g.edge_dx = d[0];
      g.edge_dy = d[1];
      g.mesh_width = (e[1] > e[0]) ? e[1] - e[0] : 1.0;
      g.mesh_height = (e[3] > e[2]) ? e[3] - e[2] : 1.0;
      g.extent_edges = (const void*)edgex;
      g.extent_nx = nx;
      g.extent_ny = ny;
    }
    if (a.export_soa) {
      /* the kernels read the store's array pointers from memory where a history ends */
      HIP_CHECK(hipMemcpyAsync(g.d_export_view, &a.p, sizeof(a.p), hipMemcpyHostToDevice,
                               g.stream));
      a.export_view = g.d_export_view;
    }
    g.tiled.slots_by_id = (pass_export && !decomposed) ? 1 : 0;
    /* (after a possible import / pending write-back above: are the arrays current now?) */
    a.export_skip_long_dead = (pass_export && !decomposed && g.soa_valid) ? 1 : 0;
    if (getenv("NEUTRAL_HIP_EXPORT_ALL")) a.export_skip_long_dead = 0; /* experiment knob */
    a.edge_dx = g.edge_dx;
    a.edge_dy = g.edge_dy;
    g.tiled.cells_per_x = (double)nx / g.mesh_width;
    g.tiled.cells_per_y = (double)ny / g.mesh_height;
  } else {
    sync_soa(); /* K1/K2 work on the SoA store in place */
    if (g.rec_owner == (const void*)particles->x) {
      drop_records();
    }
  }

  /* Arithmetic policy of this step's kernels (neutral_device.h).  Auto: start from what
   * the last step's check found; the check of THIS step's input runs on the device ahead
   * of the kernels and turns a fast attempt down if the input is outside the proven
   * range (the attempt then runs again, checked).  A padded mesh's halo cells hold
   * anything, so the density check cannot speak for it: checked. */
  if (g.checked_density != (const void*)density) {
    g.checked_density = (const void*)density;
    g.use_checked = false;
  }
  bool checked = g.arithmetic == NEUTRAL_HIP_ARITH_CHECKED || g.use_checked || pad != 0;
  bool stale_view = false;

  neutral::StepCounters hc[2];
  unsigned ctrl[16] = {0};
  unsigned long long words[kStepWords] = {0}; /* several ranks: the global step words */
  g.host_collectives = 0;
  int passes = 0;
  int same = 0;
  int attempts = 0;
  /* HIP-event times of the step's stages, ACCUMULATED over every batch of launches the
   * step needs (the first enqueue, more stream passes when the step outruns the plan,
   * the rounds of a decomposed mesh): each batch brackets itself with the same events
   * and is harvested after the wait that follows it. */
  struct StageMs {
    double kernel = 0.0, sort = 0.0, stream = 0.0, collide = 0.0, exported = 0.0;
  } stage;
  auto harvest = [&](bool with_sort) {
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, g.ev_start, g.ev_stop));
    stage.kernel += (double)ms;
    if (tiled) {
      if (with_sort) {
        HIP_CHECK(hipEventElapsedTime(&ms, g.ev_start, g.ev_sorted));
        stage.sort += (double)ms;
        HIP_CHECK(hipEventElapsedTime(&ms, g.ev_sorted, g.ev_streamed));
      } else { /* (later sorts sit inside the stream passes they serve) */
        HIP_CHECK(hipEventElapsedTime(&ms, g.ev_start, g.ev_streamed));
      }
      stage.stream += (double)ms;
      HIP_CHECK(hipEventElapsedTime(&ms, g.ev_streamed, g.ev_collected));
      stage.sort += (double)ms; /* (the collision queue's build) */
      HIP_CHECK(hipEventElapsedTime(&ms, g.ev_collected, g.ev_stop));
      stage.collide += (double)ms;
    } else {
      stage.collide += (double)ms;
    }
    HIP_CHECK(hipEventElapsedTime(&ms, g.ev_stop, g.ev_exported));
    stage.exported += (double)ms;
  };
  for (int attempt = 0;; ++attempt) {
    attempts++;
    /* is every density inside the proven range?  Asked every step, like the tables (a
     * pass over nx * ny doubles: microseconds), read with the step's counters */
    HIP_CHECK(hipMemsetAsync(g.d_check + 5, 0, sizeof(unsigned long long), g.stream));
    if (pad == 0) {
      HIP_CHECK(neutral::launch_unphysical_values(density, (long long)nx * ny, g.d_check + 5,
                                                  g.stream));
    }
    a.checked = checked ? 1 : 0;
    /* Identical tables (the shipped elastic_scatter.cs / capture.cs are) need one
     * search per energy instead of two, and both searches start from a bucketed
     * index.  The view is cached and its validity checked on the device (see
     * TableView): when the check fails the kernels of this attempt have done
     * nothing, and the step runs again with a fresh view. */
    refresh_table_view(cs_scatter_table, cs_absorb_table, stale_view, !checked);
    const TableView& v = g.tables;
    same = v.same;
    a.scatter_keys = cs_scatter_table->keys;
    a.scatter_values = cs_scatter_table->values;
    a.scatter_n = cs_scatter_table->nentries;
    a.absorb_keys = cs_absorb_table->keys;
    a.absorb_values = cs_absorb_table->values;
    a.absorb_n = cs_absorb_table->nentries;
    a.same_tables = same;
    a.scatter_index = v.ix_s.start;
    a.scatter_index_n = v.ix_s.nbuckets;
    a.scatter_index_base = v.ix_s.base;
    a.absorb_index = v.ix_a.start;
    a.absorb_index_n = v.ix_a.nbuckets;
    a.absorb_index_base = v.ix_a.base;
    a.index_shift = v.ix_s.start ? v.ix_s.shift : v.ix_a.shift;
    g.tiled.fine_index = nullptr;
    if (tiled && v.fine.start) {
      g.tiled.fine_index = v.fine.start;
      g.tiled.fine_index_n = v.fine.nbuckets;
      g.tiled.fine_index_base = v.fine.base;
      g.tiled.fine_index_shift = v.fine.shift;
    }
    if (tiled) {
      /* the tally window takes 128 KB of the 160 KB of LDS: an index that does
       * not fit next to it stays in HBM-side bisection (same brackets) */
      const size_t lds_limit = 160 * 1024 - 64;
      if (neutral::tiled_lds_bytes(a) > lds_limit) {
        a.absorb_index = nullptr;
      }
      if (neutral::tiled_lds_bytes(a) > lds_limit) {
        a.scatter_index = nullptr;
      }
    }

    HIP_CHECK(hipMemsetAsync(g.d_counters, 0, 2 * sizeof(neutral::StepCounters), g.stream));
    if (tiled) {
      /* (the pipeline's control words are set by its own kernels -- unless there is
       * nothing to launch them for: a rank that starts the step without particles) */
      HIP_CHECK(hipMemsetAsync(g.tiled.ctrl, 0, sizeof(unsigned) * 8, g.stream));
    }
    /* (a decomposed mesh has nothing to sum: every rank tallies its own cells) */
    const bool exchange = neutral::comm_nranks() > 1 && !decomposed;
    if (exchange) {
      a.tally = step_tally((size_t)nx * (size_t)ny);
      if (g.flux_tally) {
        a.flux_tally = step_flux((size_t)nx * (size_t)ny);
      }
    }
    HIP_CHECK(hipEventRecord(g.ev_start, g.stream));
    if (tiled) {
      /* Stream passes are enqueued on what the last step needed (plus one, which
       * finds nothing to do when the guess holds) without waiting in between; the
       * first step of a problem starts with two. */
      neutral::TiledPlan plan;
      plan.stream_passes = g.plan_passes > 0 ? g.plan_passes + 1 : 2;
      plan.blocks_per_cu = -1; /* the collision stage sizes itself from its queue */
      HIP_CHECK(neutral::launch_solve_tiled(a, g.tiled, g.stream, plan, 0, g.ev_sorted,
                                            g.ev_streamed, g.ev_collected, &passes));
    } else {
      HIP_CHECK(neutral::launch_solve(a, g.variant, g.stream));
    }
    HIP_CHECK(hipEventRecord(g.ev_stop, g.stream));
    if (exchange) {
      exchange_step(a, energy_deposition_tally, tiled); /* (beside the write-back below) */
    }
    if (pass_export && !decomposed) {
      /* this step's records (t.rec_out until the swap below) to the SoA arrays */
      HIP_CHECK(neutral::launch_export_records(
          g.tiled.rec_out, g.tiled.slot_of_id, a.p, a.nparticles, g.stream, a.abort_flag,
          a.export_skip_long_dead ? neutral::tiled_first_inactive(g.tiled) : nullptr));
    }
    HIP_CHECK(hipEventRecord(g.ev_exported, g.stream));

    if (exchange) {
      finish_exchange();
    }
    /* the one wait of a steady-state step: counters, the pipeline's control words, the
     * verdict on the table view and -- several ranks -- the step words, published by one
     * small kernel into pinned host memory */
    unsigned long long check[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    publish_results(tiled, exchange);
    wait_for_stream();
    fetch_results(hc, check, tiled ? ctrl : nullptr, exchange ? words : nullptr);
    stage = StageMs(); /* (an attempt that was turned down did nothing worth timing) */
    harvest(true);
    /* the device's verdict: [6] the cached view of the tables is stale, [7] a fast attempt
     * met input outside the proven range ([4] tables, [5] densities).  Either way the
     * kernels of this attempt returned at entry, and it runs again -- with a fresh view,
     * with the checked instantiation. */
    stale_view = check[6] != 0;
    const bool unproven = (check[4] | check[5]) != 0;
    g.use_checked = unproven; /* (the next step starts from this) */
    if (unproven && !g.said_checked && !g.quiet) {
      g.said_checked = true;
      fprintf(stderr,
              "libneutral_hip: %s%s%s outside [2^-100, 2^100] (a true vacuum of density 0, for "
              "instance): such steps run the kernels instantiated with IEEE-checked arithmetic, "
              "which follow the reference's C on infinities and NaNs.\n",
              check[5] ? "the density of some cells lies" : "", (check[4] && check[5]) ? " and " : "",
              check[4] ? "some cross-section table entries lie" : "");
    }
    if (check[7] != 0) {
      checked = true;
    }
    /* (several ranks take every decision that leads to another exchange together:
     * the collectives must pair up) */
    const bool turned_down = decomposed ? check[0] != 0
                                        : (exchange ? words[kWordTurnedDown] != 0 : check[0] != 0);
    if (!turned_down) {
      break;
    }
    if (attempt >= 3) {
      fprintf(stderr, "libneutral_hip: the cross-section tables keep changing under "
                      "solve_transport_2d.\n");
      exit(EXIT_FAILURE);
    }
  }

  const bool exchange = neutral::comm_nranks() > 1 && !decomposed;
  unsigned long long queue_total = exchange ? words[kWordQueued] : ctrl[2];
  if (tiled) {
    /* Finishes what is enqueued: migrants left over mean the step outran the plan (it
     * needs more stream passes than the last one did).  More passes, as many again as
     * have run; the histories they suspend get a collision stage of their own (the
     * first one's are marked done), and with several ranks their tallies an exchange
     * of their own. */
    auto finish_passes = [&]() {
      while (exchange ? (words[kWordMigrants] != 0) : (ctrl[4] != 0)) {
        neutral::TiledPlan more = {passes < 2 ? 2 : passes, -1};
        HIP_CHECK(hipEventRecord(g.ev_start, g.stream));
        HIP_CHECK(neutral::launch_solve_tiled(a, g.tiled, g.stream, more, passes, nullptr,
                                              g.ev_streamed, g.ev_collected, &passes));
        HIP_CHECK(hipEventRecord(g.ev_stop, g.stream));
        if (exchange) {
          exchange_step(a, energy_deposition_tally, tiled);
        }
        if (pass_export && !decomposed) {
          HIP_CHECK(neutral::launch_export_records(
              g.tiled.rec_out, g.tiled.slot_of_id, a.p, a.nparticles, g.stream, nullptr,
              a.export_skip_long_dead ? neutral::tiled_first_inactive(g.tiled) : nullptr));
        }
        HIP_CHECK(hipEventRecord(g.ev_exported, g.stream));
        if (exchange) {
          finish_exchange();
        }
        publish_results(true, exchange);
        wait_for_stream();
        fetch_results(hc, nullptr, ctrl, exchange ? words : nullptr);
        harvest(false);
        queue_total += exchange ? words[kWordQueued] : ctrl[2];
      }
    };
    finish_passes();
    if (decomposed) {
      /* Decomposed mesh: histories that crossed into another rank's block wait as
       * emigrants.  Rounds of: count and pack them by destination, exchange, append
       * the arrivals, go on with the step for them -- until no rank has any. */
      for (;;) {
        uint64_t waiting = ctrl[7];
        comms_allreduce_u64(&waiting, 1, COMMS_SUM);
        g.host_collectives++;
        if (waiting == 0) {
          break;
        }
        const int arrived = exchange_particles(a, g.tiled);
        a.nparticles += arrived;
        neutral::TiledPlan more = {2, -1};
        const int first = passes < 1 ? 1 : passes; /* (pass 0 would start histories over) */
        HIP_CHECK(hipEventRecord(g.ev_start, g.stream));
        HIP_CHECK(neutral::launch_solve_tiled(a, g.tiled, g.stream, more, first, nullptr,
                                              g.ev_streamed, g.ev_collected, &passes));
        HIP_CHECK(hipEventRecord(g.ev_stop, g.stream));
        HIP_CHECK(hipEventRecord(g.ev_exported, g.stream));
        publish_results(true, false);
        wait_for_stream();
        fetch_results(hc, nullptr, ctrl, nullptr);
        harvest(false);
        queue_total += ctrl[2];
        finish_passes();
      }
      /* the particles that are here now, without the holes the emigrants left; they
       * land in the other record buffer, which is where the next step looks */
      unsigned kept = 0;
      g.free_count = 0; /* (the holes are closed: nothing to reuse next step) */
      HIP_CHECK(neutral::launch_compact_records(g.tiled, a.nparticles, g.d_exchange + 192,
                                                g.stream));
      HIP_CHECK(hipMemcpyAsync(&kept, g.d_exchange + 192, sizeof(unsigned), hipMemcpyDeviceToHost,
                               g.stream));
      wait_for_stream();
      shard->count = (int)kept;
      *nlocal_particles = (int)kept;
      g.rec_count = (int)kept;
      if (pass_export) {
        HIP_CHECK(hipEventRecord(g.ev_stop, g.stream));
        HIP_CHECK(neutral::launch_export_by_slot(g.tiled.rec_in, a.p, shard->keys, (int)kept,
                                                 g.stream));
        HIP_CHECK(hipEventRecord(g.ev_exported, g.stream));
        wait_for_stream();
        float ms_slot = 0.0f;
        HIP_CHECK(hipEventElapsedTime(&ms_slot, g.ev_stop, g.ev_exported));
        stage.exported += (double)ms_slot;
      }
      g.plan_passes = (int)ctrl[5] > 0 ? (int)ctrl[5] : 1;
      g.soa_valid = !g.lazy_export;
    }
  }
  if (tiled && !decomposed) {
    /* this step's records become the next step's input */
    neutral::TiledArgs& t = g.tiled;
    neutral::ParticleRec* swap = t.rec_in;
    t.rec_in = t.rec_out;
    t.rec_out = swap;
    unsigned* swap_info = t.info_in;
    t.info_in = t.info_out;
    t.info_out = swap_info;
    unsigned* swap_id = t.id_in;
    t.id_in = t.id_out;
    t.id_out = swap_id;
    g.plan_passes = (int)ctrl[5] > 0 ? (int)ctrl[5] : 1;
    g.soa_valid = !g.lazy_export; /* eager: exported above (or by the kernels) */
    g.slots_valid = t.slots_by_id != 0;
    /* the graveyard grows by what the sort carried over a step ago; what it carried over
     * now joins next step (ctrl[8]: the first slot of the dead this step's sort found) */
    t.mirror_end = t.sort_end;
    t.sort_end = ((int)ctrl[8] <= t.sort_end) ? (int)ctrl[8] : t.sort_end;
  }

  /* sort_ms: the first sort and the queue builds (later sorts sit inside stream_ms) */
  const double ms = stage.kernel, ms_sort = stage.sort, ms_stream = stage.stream,
               ms_collide = stage.collide, ms_export = stage.exported;

  if (exchange) {
    /* event counters of all ranks: they travelled with the tally (StepWord) */
    for (int k = 0; k < 2; ++k) {
      hc[k].nprocessed = words[kWordCounters + 4 * k + 0];
      hc[k].nfacets = words[kWordCounters + 4 * k + 1];
      hc[k].ncollisions = words[kWordCounters + 4 * k + 2];
      hc[k].ncensus = words[kWordCounters + 4 * k + 3];
    }
    hc[0].nrequeued = 0;
    hc[1].nrequeued = words[kWordRequeued];
    hc[0].ncollide_passes = 0;
    hc[1].ncollide_passes = words[kWordCollidePasses];
    hc[0].nsteals = 0;
    hc[1].nsteals = words[kWordSteals];
    hc[0].aborted = 0;
    hc[1].aborted = (unsigned)words[kWordAborted];
  } else if (neutral::comm_nranks() > 1) {
    /* decomposed mesh: a handful of words over the host links, like its other exchanges */
    static_assert(sizeof(hc) % 8 == 0, "StepCounters is summed word by word");
    const unsigned aborted[2] = {hc[0].aborted, hc[1].aborted};
    comms_allreduce_u64((uint64_t*)hc, sizeof(hc) / 8, COMMS_SUM);
    hc[0].aborted = aborted[0]; /* (two 32-bit fields share a word: keep the local ones) */
    hc[1].aborted = aborted[1];
    uint64_t q = queue_total;
    comms_allreduce_u64(&q, 1, COMMS_SUM);
    queue_total = q;
    g.host_collectives += 2;
  }
  neutral::StepCounters h = hc[0];
  h.nprocessed += hc[1].nprocessed;
  h.nfacets += hc[1].nfacets;
  h.ncollisions += hc[1].ncollisions;
  h.ncensus += hc[1].ncensus;

  *facet_events += h.nfacets; /* omp3/neutral.c:202-203 */
  *collision_events += h.ncollisions;

  g.last.nprocessed = h.nprocessed;
  g.last.facets = h.nfacets;
  g.last.collisions = h.ncollisions;
  g.last.census = h.ncensus;
  g.last.kernel_ms = (double)ms;
  g.last.same_tables = same;
  g.last.variant = g.variant;
  g.last.sort_ms = (double)ms_sort;
  g.last.stream_ms = (double)ms_stream;
  g.last.collide_ms = (double)ms_collide;
  g.last.stream_facets = tiled ? hc[0].nfacets : 0;
  g.last.stream_census = tiled ? hc[0].ncensus : 0;
  g.last.suspended = queue_total;
  g.last.aborted = (uint64_t)hc[0].aborted + (uint64_t)hc[1].aborted;
  if (g.last.aborted) {
    fprintf(stderr, "libneutral_hip: warning: %llu histories exceeded the event watchdog and "
                    "were stopped.\n", (unsigned long long)g.last.aborted);
  }
  g.last.stream_passes = tiled ? (int)ctrl[5] : 0;
  g.last.requeued = tiled ? hc[1].nrequeued : 0;
  g.last.collide_passes = hc[0].ncollide_passes + hc[1].ncollide_passes;
  g.last.steals = hc[0].nsteals + hc[1].nsteals;
  g.last.host_syncs = g.host_syncs;
  g.last.stream_passes_enqueued = tiled ? passes : 0;
  g.last.tile_cells = tiled ? (1 << g.tiled.tile_shift) : 0;
  g.last.export_ms = (double)ms_export;
  g.last.checked_arithmetic = checked ? 1 : 0;
  g.last.attempts = attempts;
  g.last.host_collectives = g.host_collectives;
  g.last.exchange_ranks = exchange ? (int)words[kWordRanks] : 1;

  if (!g.quiet) {
    printf("Particles  %llu\n", (unsigned long long)h.nprocessed); /* omp3/neutral.c:205 */
    fflush(stdout);
  }
}

size_t inject_particles(const int nparticles, const int global_nx, const int local_nx,
                        const int local_ny, const int pad,
                        const double local_particle_left_off,
                        const double local_particle_bottom_off,
                        const double local_particle_width,
                        const double local_particle_height, const int x_off, const int y_off,
                        const double dt, const double* edgex, const double* edgey,
                        const double initial_energy, NeutralHipParticle** particles) {
  (void)global_nx;
  NeutralHipParticle* p = (NeutralHipParticle*)malloc(sizeof(NeutralHipParticle));
  if (!p) {
    fprintf(stderr, "Could not allocate particle array.\n"); /* omp3/neutral.c:571-573 */
    exit(EXIT_FAILURE);
  }
  /* several ranks: this rank's contiguous share of the ids 0..nparticles-1 (the
   * OpenMP static split of omp3/neutral.c:64-74 over ranks); ids stay global, so
   * every history is the one a single rank would run */
  int local = nparticles > 0 ? nparticles : 0;
  const bool decomposed = g.domain_on && (local_nx < global_nx || g.domain.px * g.domain.py > 1);
  const bool sharded = !decomposed && neutral::comm_nranks() > 1 && g.auto_shard;
  if (sharded) {
    long long first = 0, count = 0;
    comms_shard_range(local, neutral::comm_rank(), neutral::comm_nranks(), &first, &count);
    g.pid_base = (uint64_t)first;
    local = (int)count;
  }
  const size_t n = (size_t)local;
  size_t allocation = 0;
  double** f64[] = {&p->x,      &p->y,      &p->omega_x,      &p->omega_y,
                    &p->energy, &p->weight, &p->dt_to_census, &p->mfp_to_collision};
  for (double** f : f64) {
    *f = (double*)device_zalloc(sizeof(double) * n);
    allocation += sizeof(double) * n;
  }
  int** i32[] = {&p->cellx, &p->celly, &p->dead};
  for (int** f : i32) {
    *f = (int*)device_zalloc(sizeof(int) * n);
    allocation += sizeof(int) * n;
  }
  *particles = p;
  if (sharded) {
    remember_store(p, local, g.pid_base);
  }
  if (decomposed) {
    /* Decomposed mesh: the store has room for every particle of the problem (any of
     * them may pass through this rank's block) and starts with the ones the source
     * puts there.  `nparticles` and the particle box are the GLOBAL ones: every rank
     * looks at all candidates, so that each particle is what one rank alone would
     * have made of it. */
    State::Store* st = remember_store(p, 0, 0);
    st->decomposed = true;
    st->capacity = local;
    HIP_CHECK(hipMalloc((void**)&st->keys, sizeof(unsigned) * (n ? n : 1)));
    allocation += sizeof(unsigned) * n;
    run_inject_filtered(st, nparticles, local_nx, local_ny, pad, local_particle_left_off,
                        local_particle_bottom_off, local_particle_width, local_particle_height,
                        x_off, y_off, dt, edgex, edgey, initial_energy, p);
    return allocation;
  }

  run_inject(local, local_nx, local_ny, pad, local_particle_left_off,
             local_particle_bottom_off, local_particle_width, local_particle_height, x_off,
             y_off, dt, edgex, edgey, initial_energy, p);
  return allocation;
}

void validate(const int nx, const int ny, const char* params_filename, const int rank,
              double* energy_tally) {
  const size_t ncells = (size_t)nx * (size_t)ny;
  double* h_tally = (double*)malloc(sizeof(double) * (ncells ? ncells : 1));
  if (!h_tally) {
    fprintf(stderr, "Could not allocate the host tally.\n");
    exit(EXIT_FAILURE);
  }
  HIP_CHECK(hipMemcpyAsync(h_tally, energy_tally, sizeof(double) * ncells,
                           hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));

  /* serial sum in index order, omp3/neutral.c:524-527 */
  double global_energy_tally = 0.0;
  for (size_t ii = 0; ii < ncells; ++ii) {
    global_energy_tally += h_tally[ii];
  }
  free(h_tally);
  if (g.domain_on && neutral::comm_nranks() > 1) {
    /* decomposed mesh: every rank holds the tally of its own cells (omp3/neutral.c:530) */
    comms_allreduce_f64(&global_energy_tally, 1, COMMS_SUM);
  }

  if (rank != 0) {
    return;
  }
  printf("\nFinal global_energy_tally %.15e\n", global_energy_tally);

  int nresults = 0;
  char* keys = (char*)malloc(sizeof(char) * NEUTRAL_MAX_KEYS * (NEUTRAL_MAX_STR_LEN + 1));
  double* values = (double*)malloc(sizeof(double) * NEUTRAL_MAX_KEYS);
  if (!keys || !values ||
      !get_key_value_parameter(params_filename, g.tests_file, keys, values, &nresults) ||
      nresults < 1) {
    printf("Warning. Test entry was not found, could NOT validate.\n");
    fflush(stdout);
    free(keys);
    free(values);
    return;
  }
  printf("Expected %.12e, result was %.12e.\n", values[0], global_energy_tally);
  if (within_tolerance(values[0], global_energy_tally, NEUTRAL_VALIDATE_TOLERANCE)) {
    printf("PASSED validation.\n");
  } else {
    printf("FAILED validation.\n");
  }
  fflush(stdout);
  free(keys);
  free(values);
}

/* ---- 2. allocation hooks, HBM flavour ---------------------------------------- */

size_t allocate_data(double** buf, size_t len) {
  *buf = (double*)device_zalloc(sizeof(double) * len);
  return sizeof(double) * len;
}
size_t allocate_float_data(float** buf, size_t len) {
  *buf = (float*)device_zalloc(sizeof(float) * len);
  return sizeof(float) * len;
}
size_t allocate_int_data(int** buf, size_t len) {
  *buf = (int*)device_zalloc(sizeof(int) * len);
  return sizeof(int) * len;
}
size_t allocate_uint64_data(uint64_t** buf, size_t len) {
  *buf = (uint64_t*)device_zalloc(sizeof(uint64_t) * len);
  return sizeof(uint64_t) * len;
}
void allocate_host_data(double** buf, size_t len) {
  *buf = (double*)calloc(len ? len : 1, sizeof(double));
  if (!*buf) {
    fprintf(stderr, "Could not allocate host data.\n");
    exit(EXIT_FAILURE);
  }
}
void allocate_host_int_data(int** buf, size_t len) {
  *buf = (int*)calloc(len ? len : 1, sizeof(int));
  if (!*buf) {
    fprintf(stderr, "Could not allocate host data.\n");
    exit(EXIT_FAILURE);
  }
}
void deallocate_data(double* buf) { HIP_CHECK(hipFree(buf)); }
void deallocate_int_data(int* buf) { HIP_CHECK(hipFree(buf)); }
void deallocate_uint64_data(uint64_t* buf) { HIP_CHECK(hipFree(buf)); }
void deallocate_host_data(double* buf) { free(buf); }

void copy_buffer(const size_t len, double** src, double** dst, int send) {
  const hipMemcpyKind kind = send ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
  if (send) {
    if (touches_record_owner(*src, sizeof(double) * len)) sync_soa(); /* lazy export pending */
  } else {
    before_device_write(*dst, sizeof(double) * len);
  }
  HIP_CHECK(hipMemcpyAsync(*dst, *src, sizeof(double) * len, kind, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}
void copy_int_buffer(const size_t len, int** src, int** dst, int send) {
  const hipMemcpyKind kind = send ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
  if (send) {
    if (touches_record_owner(*src, sizeof(int) * len)) sync_soa(); /* lazy export pending */
  } else {
    before_device_write(*dst, sizeof(int) * len);
  }
  HIP_CHECK(hipMemcpyAsync(*dst, *src, sizeof(int) * len, kind, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}
void move_host_buffer_to_device(const size_t len, double** src, double** dst) {
  HIP_CHECK(hipMalloc((void**)dst, sizeof(double) * (len ? len : 1)));
  HIP_CHECK(hipMemcpyAsync(*dst, *src, sizeof(double) * len, hipMemcpyHostToDevice, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  free(*src);
  *src = NULL;
}

/* ---- 3. extensions ------------------------------------------------------------ */

int neutral_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    return 0;
  }
  return n;
}

int neutral_hip_set_device(int device) {
  return hipSetDevice(device) == hipSuccess ? 0 : 1;
}

void neutral_hip_set_stream(void* hip_stream) { g.stream = (hipStream_t)hip_stream; }
void neutral_hip_set_pid_base(uint64_t pid_base) { g.pid_base = pid_base; }
uint64_t neutral_hip_get_pid_base(void) { return g.pid_base; }

int neutral_hip_set_variant(int variant) {
  if (variant != NEUTRAL_HIP_VARIANT_OVER_PARTICLE &&
      variant != NEUTRAL_HIP_VARIANT_EVENT_SORTED && variant != NEUTRAL_HIP_VARIANT_TILED) {
    return 1;
  }
  g.variant = variant;
  g.variant_from_env_done = true; /* an explicit choice overrides the environment */
  return 0;
}

void neutral_hip_set_quiet(int quiet) { g.quiet = quiet; }

int neutral_hip_set_arithmetic(int mode) {
  if (mode != NEUTRAL_HIP_ARITH_AUTO && mode != NEUTRAL_HIP_ARITH_CHECKED) {
    return 1;
  }
  g.arithmetic = mode;
  g.arithmetic_from_env_done = true; /* an explicit choice overrides the environment */
  g.use_checked = false;             /* (auto mode starts over: the next step's check decides) */
  return 0;
}

void neutral_hip_set_tests_file(const char* path) {
  strncpy(g.tests_file, path, NEUTRAL_MAX_STR_LEN - 1);
  g.tests_file[NEUTRAL_MAX_STR_LEN - 1] = '\0';
}

void neutral_hip_last_step(NeutralHipStepStats* stats) { *stats = g.last; }

void neutral_hip_reinject_particles(const int nparticles, const int local_nx,
                                    const int local_ny, const int pad,
                                    const double local_particle_left_off,
                                    const double local_particle_bottom_off,
                                    const double local_particle_width,
                                    const double local_particle_height, const int x_off,
                                    const int y_off, const double dt, const double* edgex,
                                    const double* edgey, const double initial_energy,
                                    NeutralHipParticle* particles) {
  const State::Store* st = find_store(particles);
  if (st && st->decomposed) {
    run_inject_filtered(const_cast<State::Store*>(st), st->capacity, local_nx, local_ny, pad,
                        local_particle_left_off, local_particle_bottom_off, local_particle_width,
                        local_particle_height, x_off, y_off, dt, edgex, edgey, initial_energy,
                        particles);
    return;
  }
  if (st) {
    g.pid_base = st->first; /* this rank's shard, whatever count the caller names */
  }
  run_inject(st ? st->count : nparticles, local_nx, local_ny, pad, local_particle_left_off,
             local_particle_bottom_off, local_particle_width, local_particle_height, x_off,
             y_off, dt, edgex, edgey, initial_energy, particles);
}

void neutral_hip_set_lazy_export(int lazy) { g.lazy_export = lazy; }

void neutral_hip_sync_particles(NeutralHipParticle* particles) {
  (void)particles; /* at most one store has a pending write-back */
  sync_soa();
}

void neutral_hip_invalidate_particles(NeutralHipParticle* particles) {
  if (particles && g.rec_owner == (const void*)particles->x) {
    /* whatever the records hold that the arrays do not have yet goes out first, so
     * a caller that changed SOME particles keeps the others */
    sync_soa();
    drop_records();
  }
}

void neutral_hip_set_scalar_flux_tally(double* device_tally) { g.flux_tally = device_tally; }

void neutral_hip_set_auto_shard(int on) { g.auto_shard = on ? 1 : 0; }

int neutral_hip_set_decomposition(int ranks_x, int ranks_y, int global_nx, int global_ny,
                                  int* x_off, int* y_off, int* local_nx, int* local_ny) {
  const int n = neutral::comm_nranks();
  if (ranks_x < 1 || ranks_y < 1 || ranks_x * ranks_y != n || n > 64 ||
      ranks_x > global_nx || ranks_y > global_ny) {
    return 1;
  }
  g.domain.px = ranks_x;
  g.domain.py = ranks_y;
  g.domain.bx = (global_nx + ranks_x - 1) / ranks_x;
  g.domain.by = (global_ny + ranks_y - 1) / ranks_y;
  /* (every rank must own at least one column and one row of cells) */
  if (g.domain.bx * (ranks_x - 1) >= global_nx || g.domain.by * (ranks_y - 1) >= global_ny) {
    return 1;
  }
  g.domain_on = true;
  const int r = neutral::comm_rank();
  const int rx = r % ranks_x;
  const int ry = r / ranks_x;
  *x_off = rx * g.domain.bx;
  *y_off = ry * g.domain.by;
  *local_nx = (rx == ranks_x - 1) ? global_nx - *x_off : g.domain.bx;
  *local_ny = (ry == ranks_y - 1) ? global_ny - *y_off : g.domain.by;
  return 0;
}

void neutral_hip_clear_decomposition(void) {
  g.domain_on = false;
  g.domain = neutral::DomainGrid{1, 1, 0, 0};
  g.source_box_set = false;
}

void neutral_hip_set_source_box(double left, double bottom, double width, double height) {
  g.source_box[0] = left;
  g.source_box[1] = bottom;
  g.source_box[2] = width;
  g.source_box[3] = height;
  g.source_box_set = true;
}

const unsigned* neutral_hip_store_keys(const NeutralHipParticle* particles) {
  const State::Store* st = find_store(particles);
  if (st && st->decomposed && g.rec_owner == (const void*)particles->x) {
    sync_soa(); /* (lazy export: the keys move with the arrays) */
  }
  return (st && st->decomposed) ? st->keys : nullptr;
}

int neutral_hip_store_count(const NeutralHipParticle* particles) {
  const State::Store* st = find_store(particles);
  return st ? st->count : -1;
}

void neutral_hip_free_particles(NeutralHipParticle* p) {
  if (!p) {
    return;
  }
  forget_store(p);
  if (g.rec_owner == (const void*)p->x) {
    g.rec_owner = nullptr; /* pending state dies with the store */
    drop_records();
    g.soa_valid = true;
  }
  void* arrays[] = {p->x,      p->y,           p->omega_x,          p->omega_y, p->energy,
                    p->weight, p->dt_to_census, p->mfp_to_collision, p->cellx,   p->celly,
                    p->dead};
  for (void* a : arrays) {
    if (a) {
      HIP_CHECK(hipFree(a));
    }
  }
  free(p);
}

void neutral_hip_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes) {
  if (touches_record_owner(src_device, bytes)) sync_soa(); /* lazy export pending */
  HIP_CHECK(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}
void neutral_hip_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes) {
  before_device_write(dst_device, bytes);
  HIP_CHECK(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}
void neutral_hip_memset(void* dst_device, int value, size_t bytes) {
  before_device_write(dst_device, bytes);
  HIP_CHECK(hipMemsetAsync(dst_device, value, bytes, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
}

}  // extern "C"

/* probes: host arrays in, host arrays out; staging through HBM inside */
namespace {
template <typename T>
T* stage_in(const T* host, size_t n) {
  T* d = nullptr;
  HIP_CHECK(hipMalloc((void**)&d, sizeof(T) * (n ? n : 1)));
  if (host && n) {
    HIP_CHECK(hipMemcpyAsync(d, host, sizeof(T) * n, hipMemcpyHostToDevice, g.stream));
  }
  return d;
}
template <typename T>
void stage_out(T* host, T* dev, size_t n) {
  HIP_CHECK(hipMemcpyAsync(host, dev, sizeof(T) * n, hipMemcpyDeviceToHost, g.stream));
  HIP_CHECK(hipStreamSynchronize(g.stream));
  HIP_CHECK(hipFree(dev));
}
}  // namespace

extern "C" {

void neutral_hip_probe_threefry(const uint64_t* in3, uint64_t* out2, double* rn2, int n) {
  uint64_t* d_in = stage_in(in3, (size_t)3 * n);
  uint64_t* d_out = stage_in((const uint64_t*)nullptr, (size_t)2 * n);
  double* d_rn = stage_in((const double*)nullptr, (size_t)2 * n);
  HIP_CHECK(neutral::launch_probe_threefry(d_in, d_out, d_rn, n, g.stream));
  stage_out(out2, d_out, (size_t)2 * n);
  stage_out(rn2, d_rn, (size_t)2 * n);
  HIP_CHECK(hipFree(d_in));
}

void neutral_hip_probe_cs_lookup(const NeutralHipCrossSection* cs, const double* energy,
                                 double* value, int* index, int n, int use_index) {
  double* d_e = stage_in(energy, (size_t)n);
  double* d_v = stage_in((const double*)nullptr, (size_t)n);
  int* d_i = stage_in((const int*)nullptr, (size_t)n);
  ensure_scratch();
  neutral::CsIndex ix = {nullptr, 0, 0, 0};
  if (use_index) {
    ix = build_index(cs->keys, cs->nentries, g.d_index[0]);
  }
  HIP_CHECK(neutral::launch_probe_cs(cs->keys, cs->values, cs->nentries, d_e, d_v, d_i, n, ix,
                                     g.stream));
  stage_out(value, d_v, (size_t)n);
  stage_out(index, d_i, (size_t)n);
  HIP_CHECK(hipFree(d_e));
}

void neutral_hip_probe_distance_to_facet(const double* in9, double* distance, int* x_facet,
                                         int n) {
  double* d_in = stage_in(in9, (size_t)9 * n);
  double* d_d = stage_in((const double*)nullptr, (size_t)n);
  int* d_x = stage_in((const int*)nullptr, (size_t)n);
  HIP_CHECK(neutral::launch_probe_facet(d_in, d_d, d_x, n, g.stream));
  stage_out(distance, d_d, (size_t)n);
  stage_out(x_facet, d_x, (size_t)n);
  HIP_CHECK(hipFree(d_in));
}

void neutral_hip_probe_division(const double* in2, double* out2, int* plain, int n) {
  double* d_in = stage_in(in2, (size_t)2 * n);
  double* d_out = stage_in((const double*)nullptr, (size_t)2 * n);
  int* d_p = stage_in((const int*)nullptr, (size_t)n);
  HIP_CHECK(neutral::launch_probe_division(d_in, d_out, d_p, n, g.stream));
  stage_out(out2, d_out, (size_t)2 * n);
  stage_out(plain, d_p, (size_t)n);
  HIP_CHECK(hipFree(d_in));
}

void neutral_hip_probe_log(const double* x, double* out8, int n) {
  double* d_in = stage_in(x, (size_t)n);
  double* d_out = stage_in((const double*)nullptr, (size_t)8 * n);
  HIP_CHECK(neutral::launch_probe_log(d_in, d_out, n, g.stream));
  stage_out(out8, d_out, (size_t)8 * n);
  HIP_CHECK(hipFree(d_in));
}

void neutral_hip_synchronize(void) { HIP_CHECK(hipStreamSynchronize(g.stream)); }
int neutral_hip_abi_version(void) { return NEUTRAL_ABI_VERSION; }

}  // extern "C"
