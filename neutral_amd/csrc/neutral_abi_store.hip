/*
 * neutral_abi_store.hip -- particle stores, the tiled variant's workspace and the records that
 * mirror a store's SoA arrays, the cached view of the cross-section tables, the HBM flavour of
 * the reference's allocation / copy hooks, settings and probes (include/neutral_hip.h, sections
 * 2 and 3).  Host code only.
 */
#include "neutral_abi_state.h"

namespace neutral_abi {

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
  HIP_CHECK(hipMalloc((void**)&g.d_exchange, sizeof(unsigned) * 200));
  HIP_CHECK(hipMalloc((void**)&g.d_words, sizeof(unsigned long long) * kStepWords));
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
  HIP_CHECK(hipEventCreate(&g.ev_exchanged));
  HIP_CHECK(hipEventCreate(&g.ev_exchange_begins));
  HIP_CHECK(hipStreamCreateWithFlags(&g.comm_stream, hipStreamNonBlocking));
  {
    /* (lowest priority: what runs there gets the CUs the caller's stream leaves free) */
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_CHECK(hipStreamCreateWithPriority(&g.export_stream, hipStreamNonBlocking, least));
    HIP_CHECK(hipEventCreateWithFlags(&g.ev_split_done, hipEventDisableTiming));
    HIP_CHECK(hipMalloc((void**)&g.d_export_view, sizeof(neutral::ParticleView)));
    g.export_view_uploaded = false;
  }
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
      /* (slot_of_id is kept by every step, lazy or not: the kernels that place a record note
       * where -- 0.3 ms per step at 1e8 -- and the write-back on demand costs what the pass
       * inside a step costs, not twice that for reading the ids out of the records first) */
      HIP_CHECK(neutral::launch_export_records(g.tiled.rec_in, g.tiled.slot_of_id,
                                               g.rec_owner_view, g.rec_count, g.stream, nullptr,
                                               nullptr, g.final_from));
      g.final_from = (unsigned)g.tiled.sort_end; /* (the graveyard as of now: final in the arrays) */
    }
    wait_for_stream();
  }
  g.soa_valid = true;
}

/* The records no longer mirror their SoA store (it was rewritten, or the record
 * layout changes): the next tiled step imports it again. */
void drop_records() {
  g.rec_valid = false;
  g.carried_valid = false; /* (TiledArgs::carried_in belongs to the records) */
  g.suspended_share = -1.0;
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
                   t.slot_of_id, t.tile_uniform, t.carried_in, t.carried_out};
    for (void* p : old) {
      if (p) HIP_CHECK(hipFree(p));
    }
    const size_t n = (size_t)nparticles;
    const size_t nb = (size_t)(tx * ty * 4 + 2); /* (up to four reach classes per tile) */
    HIP_CHECK(hipMalloc((void**)&t.order, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.collide_queue, sizeof(unsigned) * n));
    if (t.susp_ids) HIP_CHECK(hipFree(t.susp_ids));
    g.susp_id_words = (n + 31) / 32;
    HIP_CHECK(hipMalloc((void**)&t.susp_ids, sizeof(unsigned) * g.susp_id_words));
    HIP_CHECK(hipMalloc((void**)&t.rec_in, sizeof(neutral::ParticleRec) * n));
    HIP_CHECK(hipMalloc((void**)&t.rec_out, sizeof(neutral::ParticleRec) * n));
    HIP_CHECK(hipMalloc((void**)&t.info_in, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.info_out, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.id_in, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.id_out, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.slot_of_id, sizeof(unsigned) * n));
    HIP_CHECK(hipMalloc((void**)&t.carried_in, sizeof(neutral::CarriedStart) * n));
    HIP_CHECK(hipMalloc((void**)&t.carried_out, sizeof(neutral::CarriedStart) * n));
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
  if (!t.steal) {
    /* rings' control words and CU lists of the collision stage's stealing: they belong to
     * this workspace and are reset by a kernel of the launch that uses them */
    t.steal = (neutral::StealWork*)device_zalloc(sizeof(neutral::StealWork));
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
  /* The stream kernel's tile queues: a log per tile, sized for what a tile can receive in one
   * launch -- every particle once (a source box inside one tile sends them all through its
   * neighbours), at least 4 096 places, 4 GiB of logs at most, of which only what is used is ever
   * touched (a tile that receives more overflows into the pass mechanism:
   * NeutralHipStepStats.stream_overflows).
   * Off by default (measured: at best level with the pass mechanism, slower on sparse decks --
   * DESIGN.md section 4 item 12); neutral_hip_set_stream_queues(1) or NEUTRAL_STREAM_QUEUES=1
   * turn them on. */
  {
    const char* switch_env = getenv("NEUTRAL_STREAM_QUEUES");
    const bool off = !(switch_env ? atoi(switch_env) != 0 : g.stream_queues != 0);
    size_t cap = ((size_t)nparticles + 1023) & ~(size_t)1023;
    cap = cap < 4096 ? 4096 : (cap > ((size_t)1 << 22) ? ((size_t)1 << 22) : cap);
    while ((size_t)t.ntiles * cap * sizeof(unsigned) > ((size_t)4 << 30) && cap > 1024) {
      cap >>= 1;
    }
    if (const char* force = getenv("NEUTRAL_STREAM_QUEUE_CAPACITY")) { /* (tests: overflow) */
      if (atoi(force) > 0) cap = (size_t)atoi(force);
    }
    const size_t want = off ? 0 : (size_t)t.ntiles * cap;
    if (want != g.queue_places || (int)t.ntiles != g.queue_tiles) {
      if (t.queue_entries) HIP_CHECK(hipFree(t.queue_entries));
      if (t.queue_tail) HIP_CHECK(hipFree(t.queue_tail));
      if (t.queue_head) HIP_CHECK(hipFree(t.queue_head));
      t.queue_entries = t.queue_tail = t.queue_head = nullptr;
      if (want) {
        HIP_CHECK(hipMalloc((void**)&t.queue_entries, sizeof(unsigned) * want));
        HIP_CHECK(hipMemsetAsync(t.queue_entries, 0xFF, sizeof(unsigned) * want, g.stream));
        t.queue_tail = (unsigned*)device_zalloc(sizeof(unsigned) * (size_t)t.ntiles);
        t.queue_head = (unsigned*)device_zalloc(sizeof(unsigned) * (size_t)t.ntiles);
      }
      g.queue_places = want;
      g.queue_tiles = t.ntiles;
    }
    t.queue_capacity = (unsigned)cap;
  }
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
    g.carried_valid = false; /* (the records' carried cross sections were looked up in other tables) */
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
    /* (NEUTRAL_NO_FINE_INDEX=1: experiments that need the collision stage's workgroups small in
     * LDS -- 17 KB instead of 34 -- to sit beside another kernel's; read once per view) */
    const char* no_fine = getenv("NEUTRAL_NO_FINE_INDEX");
    if (v.same && v.ix_s.start && g.variant == NEUTRAL_HIP_VARIANT_TILED && !(no_fine && atoi(no_fine) != 0)) {
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

}  // namespace neutral_abi

using namespace neutral_abi;

extern "C" {

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

void neutral_hip_set_stream_queues(int on) { g.stream_queues = on ? 1 : 0; }

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

void neutral_hip_probe_scatter(const double* in4, double* out10, int n) {
  double* d_in = stage_in(in4, (size_t)4 * n);
  double* d_out = stage_in((const double*)nullptr, (size_t)10 * n);
  HIP_CHECK(neutral::launch_probe_scatter(d_in, d_out, n, g.stream));
  stage_out(out10, d_out, (size_t)10 * n);
  HIP_CHECK(hipFree(d_in));
}

void neutral_hip_synchronize(void) { HIP_CHECK(hipStreamSynchronize(g.stream)); }
int neutral_hip_abi_version(void) { return NEUTRAL_ABI_VERSION; }

}  // extern "C"
