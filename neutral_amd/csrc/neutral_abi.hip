/*
 * neutral_abi.hip -- the C ABI of libneutral_hip.so (include/neutral_hip.h):
 * the three functions of the reference's neutral_interface.h, the HBM flavour
 * of the allocation hooks, and the extension entry points.  Host code only;
 * kernels live in neutral_kernels.hip.
 */
#include "neutral_abi_state.h"

using namespace neutral_abi;

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
  g.exchange_rounds = 0;
  g.emigrants = 0;
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
  a.slot_info = nullptr;
  a.tiles_x = 0;
  a.tile_shift = 4;
  a.susp = nullptr;
  a.carried = nullptr;
  a.steal = nullptr;
  a.occupancy_rows = 0;
  /* the launches' tuning: read from the environment and the runtime once per store -- here for
   * a store stepped for the first time (below, where its records are imported), or on the
   * library's first step */
  if (!g.tuning_read) {
    g.tuning = neutral::launch_tuning_from_env();
    g.tuning_read = true;
  }
  /* Default (eager) mode: the SoA arrays are current when the call returns.  One
   * export pass at the end of the step does that (5.9 ms at 1e8 particles); letting
   * every kernel that ends a history store it to the arrays itself -- eleven
   * scattered 8-byte stores per history -- cost 18 ms (profiles/r02: fused export). */
  const bool pass_export = tiled && !g.lazy_export;
  a.export_skip_long_dead = 0;
  a.export_view = nullptr;
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
      g.tuning = neutral::launch_tuning_from_env(); /* (once per store) */
      if (decomposed) {
        HIP_CHECK(neutral::launch_import_by_slot(a.p, shard->keys, g.tiled, x_off, y_off,
                                                 a.nparticles, g.stream));
      } else {
        HIP_CHECK(neutral::launch_import_records(a.p, g.tiled.rec_in, g.tiled.info_in,
                                                 g.tiled.slot_of_id, g.tiled.id_in, g.tiled.tiles_x,
                                                 g.tiled.tile_shift, x_off, y_off, a.nparticles,
                                                 g.stream));
      }
      g.tiled.sort_end = a.nparticles; /* (no graveyard yet) */
      g.tiled.mirror_end = a.nparticles;
      g.final_from = 0xFFFFFFFFu;
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
    g.tiled.slots_by_id = decomposed ? 0 : 1; /* (a decomposed store keeps id_out[slot] instead) */
    /* (after a possible import / pending write-back above: are the arrays current now?) */
    a.export_skip_long_dead = (pass_export && !decomposed && g.soa_valid) ? 1 : 0;
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

  a.steal_min = g.tuning.steal_min;
  a.steal_delay = g.tuning.steal_delay;
  a.share_weight = g.tuning.share_weight;
  a.weighted_share_min = g.tuning.weighted_share_min;
  a.compute_units = g.tuning.compute_units;
  a.max_blocks = tiled ? g.tuning.max_blocks : 0;

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
    double kernel = 0.0, sort = 0.0, stream = 0.0, collide = 0.0, exported = 0.0, exchange = 0.0;
  } stage;
  uint64_t local_nprocessed = 0; /* (this rank's own, before the ranks' counters are summed) */
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
    if (neutral::comm_nranks() > 1 && !decomposed) {
      HIP_CHECK(hipEventElapsedTime(&ms, g.ev_exchange_begins, g.ev_exchanged));
      stage.exchange += (double)ms;
    }
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
      if (neutral::tiled_lds_bytes(a, g.tiled) > lds_limit) {
        a.absorb_index = nullptr;
      }
      if (neutral::tiled_lds_bytes(a, g.tiled) > lds_limit) {
        a.scatter_index = nullptr;
      }
    }

    if (tiled && neutral::tiled_uses_carried(a, g.tiled) && !g.carried_valid) {
      /* the stream kernel starts histories from the cross section carried with each record:
       * looked up here for a store just imported, or after the table view was rebuilt (an
       * attempt that is turned down for a stale view comes back through here) */
      HIP_CHECK(neutral::launch_refresh_micro(a, g.tiled, g.stream));
      g.carried_valid = true;
    }
    HIP_CHECK(hipMemsetAsync(g.d_counters, 0, 2 * sizeof(neutral::StepCounters), g.stream));
    if (tiled) {
      /* (the pipeline's control words are set by its own kernels -- unless there is
       * nothing to launch them for: a rank that starts the step without particles) */
      HIP_CHECK(hipMemsetAsync(g.tiled.ctrl, 0, sizeof(unsigned) * 16, g.stream));
    }
    /* (a decomposed mesh has nothing to sum: every rank tallies its own cells) */
    const bool exchange = neutral::comm_nranks() > 1 && !decomposed;
    if (exchange) {
      a.tally = step_tally((size_t)nx * (size_t)ny);
      if (g.flux_tally) {
        a.flux_tally = step_flux((size_t)nx * (size_t)ny);
      }
    }
    /* The write-back in two parts, when under half of the particles went to the collision stage
     * last step (csp: a tenth): the pass over the ids of everybody else runs on a stream of lowest
     * priority BESIDE the collision stage instead of after it (their records are final when the
     * stream kernel is through), and the collision stage writes the final state of the histories
     * it ends to the arrays itself (eleven scattered stores each, behind its arithmetic).
     * NEUTRAL_SPLIT_EXPORT=0: the one pass. */
    neutral::SplitExport split = {};
    split.on = false;
    if (pass_export && !decomposed && g.suspended_share >= 0.0 && g.suspended_share < 0.5) {
      const char* off = getenv("NEUTRAL_SPLIT_EXPORT");
      split.on = !(off && atoi(off) == 0);
    }
    a.export_view = nullptr;
    g.tiled.mark_suspended = 0;
    if (split.on) {
      /* (the stepped store's eleven array pointers: uploaded when they change, not every step --
       * a copy out of pageable host memory is a staging kernel of 70-130 us in the kernel trace) */
      if (memcmp(&g.h_export_view, &a.p, sizeof(a.p)) != 0 || !g.export_view_uploaded) {
        g.h_export_view = a.p;
        HIP_CHECK(hipMemcpyAsync(g.d_export_view, &g.h_export_view, sizeof(a.p), hipMemcpyHostToDevice,
                                 g.stream));
        g.export_view_uploaded = true;
      }
      HIP_CHECK(hipMemsetAsync(g.tiled.susp_ids, 0, sizeof(unsigned) * g.susp_id_words, g.stream));
      a.export_view = g.d_export_view;
      g.tiled.mark_suspended = 1;
      split.side = g.export_stream;
      split.done = g.ev_split_done;
      split.p = a.p;
      split.skip_long_dead = a.export_skip_long_dead;

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
                                            g.ev_streamed, g.ev_collected, &passes, &split));
    } else {
      HIP_CHECK(neutral::launch_solve(a, g.variant, g.stream));
    }
    HIP_CHECK(hipEventRecord(g.ev_stop, g.stream));
    if (exchange) {
      exchange_step(a, energy_deposition_tally, tiled); /* (beside the write-back below) */
    }
    if (split.on) {
      /* (the pass beside the collision stage: the caller's stream goes on when it is through) */
      HIP_CHECK(neutral::launch_split_export(a, g.tiled, split, g.ev_collected));
      HIP_CHECK(hipStreamWaitEvent(g.stream, g.ev_split_done, 0));
    } else if (pass_export && !decomposed) {
      /* this step's records (t.rec_out until the swap below) to the SoA arrays */
      HIP_CHECK(neutral::launch_export_records(
          g.tiled.rec_out, g.tiled.slot_of_id, a.p, a.nparticles, g.stream, a.abort_flag,
          a.export_skip_long_dead ? neutral::tiled_first_inactive(g.tiled) : nullptr, 0xFFFFFFFFu,
          nullptr, 0, a.export_skip_long_dead != 0));
    }
    HIP_CHECK(hipEventRecord(g.ev_exported, g.stream));
    /* (whatever else this step enqueues -- more passes for a step that outran its plan -- is
     * followed by the one pass over everybody) */
    a.export_view = nullptr;
    g.tiled.mark_suspended = 0;

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
    local_nprocessed = hc[0].nprocessed + hc[1].nprocessed;
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
              a.export_skip_long_dead ? neutral::tiled_first_inactive(g.tiled) : nullptr, 0xFFFFFFFFu,
              nullptr, 0, a.export_skip_long_dead != 0));
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
    neutral::CarriedStart* swap_carried = t.carried_in;
    t.carried_in = t.carried_out;
    t.carried_out = swap_carried;
    if (!t.carried) {
      g.carried_valid = false; /* (a step that looked up and drew itself kept none of it) */
    }
    g.plan_passes = (int)ctrl[5] > 0 ? (int)ctrl[5] : 1;
    g.soa_valid = !g.lazy_export; /* eager: exported above (or by the kernels) */
    g.suspended_share = (double)queue_total /
                        ((double)(a.nparticles > 0 ? a.nparticles : 1) * (double)neutral::comm_nranks());
    /* the graveyard grows by what the sort carried over a step ago; what it carried over
     * now joins next step (ctrl[8]: the first slot of the dead this step's sort found) */
    t.mirror_end = t.sort_end;
    t.sort_end = ((int)ctrl[8] <= t.sort_end) ? (int)ctrl[8] : t.sort_end;
    if (g.soa_valid) {
      g.final_from = (unsigned)t.sort_end; /* (the arrays are current: so is the graveyard in them) */
    }
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
    hc[0].steal_refused = 0;
    hc[1].steal_refused = words[kWordStealsRefused];
    hc[0].nweighted = 0;
    hc[1].nweighted = words[kWordWeightedWaves];
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
    fprintf(stderr, "libneutral_hip: warning: %llu histories exceeded the event watchdog or were "
                    "dropped by a consistency check of the stream kernel's tile queues, and were "
                    "stopped: the step's results are incomplete.\n", (unsigned long long)g.last.aborted);
  }
  g.last.stream_passes = tiled ? (int)ctrl[5] : 0;
  g.last.requeued = tiled ? hc[1].nrequeued : 0;
  g.last.collide_passes = hc[0].ncollide_passes + hc[1].ncollide_passes;
  g.last.steals = hc[0].nsteals + hc[1].nsteals;
  g.last.steals_refused = hc[0].steal_refused + hc[1].steal_refused;
  g.last.weighted_waves = hc[0].nweighted + hc[1].nweighted;
  /* (this rank's own launches: the clocks are not summed over ranks) */
  g.last.stream_clock_ghz = hc[0].clock_100mhz_ticks
                                ? (double)hc[0].clock_shader_ticks / ((double)hc[0].clock_100mhz_ticks * 10.0)
                                : 0.0;
  g.last.collide_clock_ghz = hc[1].clock_100mhz_ticks
                                 ? (double)hc[1].clock_shader_ticks / ((double)hc[1].clock_100mhz_ticks * 10.0)
                                 : 0.0;
  g.last.stream_hops = tiled ? ctrl[10] : 0;
  g.last.stream_overflows = tiled ? ctrl[11] : 0;
  g.last.stream_batches = tiled ? ctrl[12] : 0;
  g.last.stream_idle_polls = tiled ? ctrl[13] : 0;
  g.last.local_nprocessed = local_nprocessed;
  g.last.exchange_ms = stage.exchange;
  g.last.exchange_rounds = g.exchange_rounds;
  g.last.emigrants = g.emigrants;
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

}  // extern "C"
