/*
 * neutral_abi_state.h -- what the translation units of the C ABI share: the library's state
 * (one per process: a process drives one GPU), the types it is made of, and the helpers of one
 * unit that another calls.
 *   neutral_abi.hip           the three functions of the reference's neutral_interface.h
 *   neutral_abi_store.hip     particle stores, the tiled workspace and its record mirror, the
 *                             cached view of the cs tables, the HBM allocation / copy hooks,
 *                             settings, probes
 *   neutral_abi_exchange.hip  what crosses to the host or to other ranks at the end of a batch
 *                             of launches: the published results, the tally exchange, the
 *                             particle exchange of a decomposed mesh
 */
#ifndef NEUTRAL_AMD_ABI_STATE_H
#define NEUTRAL_AMD_ABI_STATE_H

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

#define NEUTRAL_ABI_VERSION 10 /* 10: probe_scatter; 9: NeutralHipStepStats grew weighted_waves, stream_clock_ghz, collide_clock_ghz; 8: NeutralHipStepStats grew steals_refused, stream_hops, stream_overflows; 7: NeutralHipStepStats grew steals; 6: set_arithmetic, NeutralHipStepStats grew checked_arithmetic, attempts, host_collectives, exchange_ranks; 5: NeutralHipStepStats grew export_ms; 2: probe_division, NeutralHipStepStats grew requeued + collide_passes; 3: probe_log;
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


namespace neutral_abi {


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
  unsigned long long words[20]; /* kStepWords */
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
  hipEvent_t ev_exchange_begins = nullptr; /* (both on comm_stream: NeutralHipStepStats.exchange_ms) */
  TableView tables;
  /* workspace of the tiled variant, grown on demand */
  neutral::TiledArgs tiled = {};
  /* which particle store the records mirror, and which copy is current */
  const void* rec_owner = nullptr; /* particles->x of the mirrored SoA store */
  neutral::ParticleView rec_owner_view = {}; /* its arrays, for the write-back */
  int rec_count = 0;
  bool rec_valid = false;          /* records hold the current state */
  bool carried_valid = false;      /* ... and TiledArgs::carried_in the cross section of every live
                                      record's energy in the tables of the cached view */
  bool soa_valid = true;           /* SoA arrays hold the current state */
  int lazy_export = 0;
  /* what the last step of this record store needed: the next step is enqueued on that
   * assumption, without waiting for the device in between (0 / -1: nothing known) */
  int plan_passes = 0;
  unsigned final_from = 0xFFFFFFFFu; /* first slot of the graveyard (TiledArgs::sort_end) when the
                                      SoA arrays were last current: the records from there on
                                      are dead for good and the arrays have their final state */
  int host_syncs = 0;              /* waits for the device inside the current call */
  int exchange_rounds = 0;         /* decomposed mesh: rounds of the particle exchange in the current call */
  unsigned long long emigrants = 0; /* ... histories this rank sent away in it */
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
  /* the write-back split in two (neutral_kernels.h: SplitExport): the stream its first part runs
   * on beside the collision stage, its event, and what share of the particles the last step of
   * this record store handed to the collision stage (< 0: not known; the split pays below a half) */
  hipStream_t export_stream = nullptr;
  hipEvent_t ev_split_done = nullptr;
  neutral::ParticleView* d_export_view = nullptr; /* the stepped store's array pointers, for the */
  neutral::ParticleView h_export_view = {};        /* collision stage's own write-back */
  bool export_view_uploaded = false;
  double suspended_share = -1.0;
  size_t susp_id_words = 0;
  neutral::LaunchTuning tuning = {}; /* read once per store (neutral_kernels.h) */
  bool tuning_read = false;
  int stream_queues = 0;   /* neutral_hip_set_stream_queues: the stream kernel's tile queues are in use */
  size_t queue_places = 0; /* ... places allocated, for how many tiles */
  int queue_tiles = 0;
};

extern State g;

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
  kWordStealsRefused = 16, /* waves that found their CU list overfull and stole nothing */
  kWordWeightedWaves = 17, /* waves of the collision stage that were dealt a weighted share */
  kStepWords = 20,      /* (the last two spare) */
};

/* ---- neutral_abi_store.hip ---- */
void ensure_scratch();
void read_variant_env();
void wait_for_stream(); /* every wait for the device goes through here: NeutralHipStepStats.host_syncs */
void* device_zalloc(size_t bytes);
void sync_soa();
void drop_records();
void ensure_tiled_workspace(int nx, int ny, int nparticles_now, int capacity);
void before_device_write(const void* dst, size_t bytes);
void refresh_table_view(const NeutralHipCrossSection* cs_s, const NeutralHipCrossSection* cs_a,
                        bool force, bool fast_arithmetic);
neutral::ParticleView view_of(const NeutralHipParticle* p);
const State::Store* find_store(const NeutralHipParticle* p);
State::Store* remember_store(const NeutralHipParticle* p, int count, uint64_t first);
void forget_store(const NeutralHipParticle* p);
double* step_flux(size_t ncells);
double* step_tally(size_t ncells);
void run_inject(const int nparticles, const int local_nx, const int local_ny, const int pad,
                const double left_off, const double bottom_off, const double width,
                const double height, const int x_off, const int y_off, const double dt,
                const double* edgex, const double* edgey, const double initial_energy,
                const NeutralHipParticle* particles);
void run_inject_filtered(State::Store* st, const int nparticles, const int local_nx,
                         const int local_ny, const int pad, const double left_off,
                         const double bottom_off, const double width, const double height,
                         const int x_off, const int y_off, const double dt, const double* edgex,
                         const double* edgey, const double initial_energy,
                         const NeutralHipParticle* particles);

/* ---- neutral_abi_exchange.hip ---- */
void exchange_step(const neutral::SolveArgs& a, double* tally, bool tiled);
void finish_exchange();
void publish_results(bool tiled, bool with_words);
void fetch_results(neutral::StepCounters* hc, unsigned long long* check, unsigned* ctrl,
                   unsigned long long* words);
int exchange_particles(const neutral::SolveArgs& a, neutral::TiledArgs& t);

}  // namespace neutral_abi
#endif
