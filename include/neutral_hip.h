/*
 * neutral_hip.h -- C ABI of libneutral_hip.so, the MI355X (gfx950) kernel set
 * for UoB-HPC/neutral's over-particle transport path.
 *
 * The library is a drop-in "kernel set" in the reference's sense (reference
 * Makefile:2,83: KERNELS=<dir> selects one implementation of the three
 * functions of neutral_interface.h).  Section 1 declares exactly those three
 * symbols with the reference's signatures; section 2 declares the HBM flavour
 * of the parent project's allocation hooks, through which the unchanged
 * reference loader (neutral_data.c) places its buffers; section 3 holds the
 * extensions that have no reference counterpart (device/stream selection,
 * particle shards for multi-GPU, kernel variant, per-step statistics).
 *
 * Plain C types only: pointers marked [device] are HBM addresses (hipMalloc or
 * any allocator handing out device memory, e.g. a torch CUDA tensor's
 * data_ptr); all other pointers are host addresses.  All entry points are
 * synchronous on return unless stated otherwise.  Fatal errors (HIP failures,
 * allocation failures) print to stderr and exit(EXIT_FAILURE), the behaviour
 * of the reference's TERMINATE sites (omp3/neutral.c:572).
 */
#ifndef NEUTRAL_HIP_H
#define NEUTRAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data model ------------------------------------------------------------
 * Layout-identical to the reference's types, so a reference translation unit
 * compiled with -DSoA and this library agree on every struct they exchange. */

/* = CrossSection, neutral_data.h:38-43 */
typedef struct {
  double* keys;   /* [device] nentries, strictly increasing, eV */
  double* values; /* [device] nentries, barns */
  int nentries;
} NeutralHipCrossSection;

/* = Particle under -DSoA, neutral_data.h:45-61: a host struct of device arrays */
typedef struct {
  double* x;                /* [device] */
  double* y;                /* [device] */
  double* omega_x;          /* [device] */
  double* omega_y;          /* [device] */
  double* energy;           /* [device] */
  double* weight;           /* [device] */
  double* dt_to_census;     /* [device] */
  double* mfp_to_collision; /* [device] */
  int* cellx;               /* [device] */
  int* celly;               /* [device] */
  int* dead;                /* [device] sticky death flag (omp3/neutral.c:91,245) */
} NeutralHipParticle;

/* ---- 1. the reference kernel interface (neutral_interface.h:11-36) ---------- */

/* Replaces <KERNELS>/neutral.c:solve_transport_2d (omp3/neutral.c:19-40,
 * neutral_interface.h:11-20).  Advances every live local particle by one
 * timestep of length dt: re-samples its distance to collision with the
 * Threefry2x64-20 stream (key = {particle id, master_key}, counter from 0),
 * then processes collision / facet / census events until census or death,
 * adding path-length heating into energy_deposition_tally with f64 atomics.
 *   nx, ny                  local mesh extent without padding
 *   global_nx, global_ny    global mesh extent (reflective outer boundary)
 *   master_key              the timestep number tt (main.c:103)
 *   pad, x_off, y_off       halo depth and rank offsets (0 in the reference)
 *   ntotal_particles        global particle count: tallies are scaled by 1/N
 *   nlocal_particles        [host, in] particles in this store; 0 -> prints
 *                           "Out of particles" and returns
 *   neighbours, edgedx, edgedy, reduce_array0..2   accepted, unused
 *   particles               host struct of [device] arrays from inject_particles
 *   density                 [device] ny*nx, row-major (celly*nx + cellx)
 *   edgex, edgey            [device] nx+1 / ny+1 edge coordinates
 *   cs_*_table              host structs of [device] arrays
 *   energy_deposition_tally [device] ny*nx, accumulated (never zeroed here)
 *   facet_events, collision_events  [host] incremented (omp3/neutral.c:202-203)
 * Prints "Particles  <n processed>" like omp3/neutral.c:205 unless silenced. */
void solve_transport_2d(
    const int nx, const int ny, const int global_nx, const int global_ny,
    const uint64_t master_key, const int pad, const int x_off, const int y_off,
    const double dt, const int ntotal_particles, int* nlocal_particles,
    const int* neighbours, NeutralHipParticle* particles, const double* density,
    const double* edgex, const double* edgey, const double* edgedx,
    const double* edgedy, NeutralHipCrossSection* cs_scatter_table,
    NeutralHipCrossSection* cs_absorb_table, double* energy_deposition_tally,
    uint64_t* reduce_array0, uint64_t* reduce_array1, uint64_t* reduce_array2,
    uint64_t* facet_events, uint64_t* collision_events);

/* Replaces inject_particles (omp3/neutral.c:560-630, neutral_interface.h:23-31).
 * Allocates *particles (host struct + eleven [device] arrays of nparticles
 * elements) and fills particles 0..nparticles-1: position uniform in the
 * source box from stream (id, 0, ctr 0), direction from stream (id, 0, ctr 1),
 * energy = initial_energy, weight 1, dt_to_census = dt, alive.  edgex/edgey are
 * [device].  Returns the number of bytes allocated. */
size_t inject_particles(const int nparticles, const int global_nx,
                        const int local_nx, const int local_ny, const int pad,
                        const double local_particle_left_off,
                        const double local_particle_bottom_off,
                        const double local_particle_width,
                        const double local_particle_height, const int x_off,
                        const int y_off, const double dt, const double* edgex,
                        const double* edgey, const double initial_energy,
                        NeutralHipParticle** particles);

/* Replaces validate (omp3/neutral.c:520-557, neutral_interface.h:35-36): sums
 * the [device] tally in index order on the host, prints
 * "Final global_energy_tally %.15e", looks `params_filename` up in the tests
 * file (default "problems/neutral.tests", neutral_data.h:33) and prints
 * PASSED/FAILED at relative tolerance 1e-3 (neutral_data.h:27). */
void validate(const int nx, const int ny, const char* params_filename,
              const int rank, double* energy_tally);

/* ---- 2. allocation hooks, HBM flavour ---------------------------------------
 * The reference never allocates directly: neutral_data.c:97-105,168-169 and
 * the parent project's mesh/shared-data set-up call these hooks, and the object
 * linked in decides the memory space.  This flavour returns zero-filled HBM
 * (hipMalloc) for the allocate_* family and host memory for allocate_host_*.
 * Each allocate_* returns the bytes allocated. */
size_t allocate_data(double** buf, size_t len);
size_t allocate_float_data(float** buf, size_t len);
size_t allocate_int_data(int** buf, size_t len);
size_t allocate_uint64_data(uint64_t** buf, size_t len);
void allocate_host_data(double** buf, size_t len);
void allocate_host_int_data(int** buf, size_t len);
void deallocate_data(double* buf);
void deallocate_int_data(int* buf);
void deallocate_uint64_data(uint64_t* buf);
void deallocate_host_data(double* buf);
/* send = 1 (RECV): *src [device] -> *dst [host]; send = 0 (SEND): host -> device
 * (neutral_data.c:59-62 reads four edge scalars back with RECV) */
void copy_buffer(const size_t len, double** src, double** dst, int send);
void copy_int_buffer(const size_t len, int** src, int** dst, int send);
/* uploads the host buffer *src into a new [device] buffer *dst, frees *src
 * (neutral_data.c:168-169) */
void move_host_buffer_to_device(const size_t len, double** src, double** dst);

/* ---- 3. extensions (no reference counterpart) -------------------------------- */

enum {
  NEUTRAL_HIP_VARIANT_OVER_PARTICLE = 0, /* one lane owns a history */
  NEUTRAL_HIP_VARIANT_EVENT_SORTED = 1,  /* lanes regrouped by next event */
  NEUTRAL_HIP_VARIANT_TILED = 2          /* tile-sorted streaming with the tally tile in
                                            LDS, then the event-regrouped collision kernel
                                            (default) */
};

typedef struct {
  uint64_t nprocessed;  /* live particles advanced ("Particles  N") */
  uint64_t facets;      /* facet events of the last step */
  uint64_t collisions;  /* collision events of the last step */
  uint64_t census;      /* histories of the last step that ended in a census event */
  double kernel_ms;     /* HIP-event time of the step's kernels on their stream */
  int same_tables;      /* 1 when both cs tables held identical data */
  int variant;          /* kernel variant that ran */
  /* tiled variant only (0 otherwise): HIP-event time of its three stages, the
   * events the streaming kernel handled, and the histories it handed to the
   * collision kernel */
  double sort_ms;
  double stream_ms;
  double collide_ms;
  uint64_t stream_facets;
  uint64_t stream_census;
  uint64_t suspended;
  uint64_t aborted;     /* histories stopped by the event watchdog (2^27 events in one
                           timestep; 0 for every sane input) */
  int stream_passes;    /* streaming passes the step took (1 unless particles outran
                           the LDS tally window and migrated to another tile) */
  uint64_t requeued;    /* tiled variant: times the collision stage put a history back in
                           its wave's ring at the end of a time slice (0 when every
                           wave's share of the collision queue fitted its lanes) */
  uint64_t collide_passes; /* wave-level collision passes of the event-regrouped kernel
                           (variants 1, 2): collisions / (64 * collide_passes) is the
                           lane occupancy of its collision passes */
  int host_syncs;       /* times the call waited for the device (1 for a steady-state
                           step of the tiled variant: the read-back of the counters) */
  int stream_passes_enqueued; /* tiled variant: stream passes enqueued (the last step's
                           count plus one when nothing was waited for in between) */
  int tile_cells;       /* tiled variant: tile edge chosen for the problem (16..128 cells) */
  double export_ms;     /* tiled variant, default mode: HIP-event time of the write-back of
                           the records to the SoA arrays (not part of kernel_ms) */
  int checked_arithmetic; /* 1: the step ran the kernels instantiated with IEEE-checked
                             arithmetic (an input lay outside the fast sequences' proven
                             range: neutral_hip_set_arithmetic below); 0: the fast ones */
  int attempts;          /* times the step's kernels were enqueued (1 in steady state; +1
                            when the device-side check turned the attempt down: a table
                            rewritten in place, input outside the proven range) */
  int host_collectives;  /* several ranks: collectives over the ranks' host links that the
                            step made of its own, beside the exchange itself (which is RCCL on
                            the kernels' stream, or staged through those links as a fallback).
                            0 in steady state: event counters and the flags the ranks act on
                            together travel with the tally, on the device.  A decomposed mesh
                            counts the rounds of its particle exchange here */
  int exchange_ranks;    /* ranks the last tally exchange summed over (1: no exchange) */
  uint64_t steals;       /* tiled variant: times a wave of the collision stage that had emptied
                            its ring took half of what waited in the ring of a wave of its CU */
  uint64_t steals_refused; /* ... waves that would have taken but did not, because the key
                            that tells them who shares their CU (read from the hardware)
                            collected more than twice the waves a CU holds in this launch
                            (workgroups placed late enter themselves on a CU whose first ones
                            have left: that much is expected): the launch then steals nothing
                            (0 on an MI355X) */
  uint64_t stream_hops;  /* tiled variant: histories that left the tally window of the tile they
                            were streaming under with far to go and were handed, INSIDE the stream
                            kernel, to the queue of the tile they had reached (no further pass) */
  uint64_t stream_overflows; /* ... and those that found that tile's queue full and waited for
                            another pass of the stream stage instead (0 unless a tile receives
                            more than its queue holds in one launch) */
  uint64_t stream_batches; /* ... claims of a workgroup on a tile's queue (stream_hops / this =
                            histories a workgroup streams under one window placement) */
  uint64_t stream_idle_polls; /* ... times a workgroup looked for work and found none while
                            histories were still in flight elsewhere */
  uint64_t local_nprocessed; /* several ranks: live particles THIS rank advanced (nprocessed is
                            the sum over the ranks then) */
  double exchange_ms;    /* several ranks: HIP-event time of the step's tally exchange (pack, the
                            two all-reduces, the add into the caller's mesh) on the library's own
                            stream, beside the write-back; 0 with one rank */
  int exchange_rounds;   /* decomposed mesh: rounds of the particle exchange between the ranks'
                            blocks the step took (each: count, pack, exchange, append, more passes) */
  uint64_t emigrants;    /* decomposed mesh: histories THIS rank sent to other ranks' blocks */
  uint64_t weighted_waves; /* tiled variant: waves of the collision stage that were dealt a share of
                            the queue in proportion to what a wave is served (5 : 1 : 1 : 1 over
                            the four waves of a SIMD; 0: equal shares -- a small queue, a partial
                            grid) */
  double stream_clock_ghz;  /* tiled variant: the shader clock the stream kernel and the collision */
  double collide_clock_ghz; /* stage ran at, measured by one wave of every launch over its own life
                               (shader-clock ticks per tick of the constant 100-MHz clock); 0: not
                               measured.  The chip does not hold its nominal 2.4 GHz under every load */
} NeutralHipStepStats;

/* Number of visible devices (does not initialise a device context). */
int neutral_hip_device_count(void);
/* Selects the device for all following calls of this process. Returns 0 on success. */
int neutral_hip_set_device(int device);
/* Stream for all kernels and copies (a hipStream_t; NULL = the null stream). */
void neutral_hip_set_stream(void* hip_stream);
/* Particle shard: local particle i carries the global id pid_base + i as its
 * RNG key, in inject_particles and solve_transport_2d alike.
 * Default 0 = the reference's numbering. */
void neutral_hip_set_pid_base(uint64_t pid_base);
uint64_t neutral_hip_get_pid_base(void);
/* Kernel variant for solve_transport_2d (NEUTRAL_HIP_VARIANT_*); also read once
 * from the environment variable NEUTRAL_HIP_VARIANT.  Returns 0 on success. */
int neutral_hip_set_variant(int variant);
/* quiet != 0 suppresses the per-step "Particles  N" line. */
void neutral_hip_set_quiet(int quiet);
/* Path of the known-answer file used by validate(). */
void neutral_hip_set_tests_file(const char* path);
/* Statistics of the most recent solve_transport_2d call. */
void neutral_hip_last_step(NeutralHipStepStats* stats);
/* Arithmetic of the event bodies.  Divisions, square roots and the logarithm exist in
 * two instantiations of every history kernel that deliver the same bits wherever both
 * are defined: bare operation sequences, proven exact when every density of the mesh
 * and every key and value of the cross-section tables lies in [2^-100, 2^100], and
 * IEEE-checked ones that accept whatever the reference's C accepts (a true-vacuum cell
 * of density 0 runs on 1/0 = inf there, omp3/neutral.c:127-146,231).
 * NEUTRAL_HIP_ARITH_AUTO (default): decided per step ON THE DEVICE from that step's
 * density mesh and tables -- the fast kernels return at entry when the input is outside
 * the proven range and the step runs checked; nothing to rebuild, nothing to configure.
 * NEUTRAL_HIP_ARITH_CHECKED: always the checked kernels (also: environment variable
 * NEUTRAL_HIP_ARITH=checked).  Returns 0 on success. */
#define NEUTRAL_HIP_ARITH_AUTO 0
#define NEUTRAL_HIP_ARITH_CHECKED 1
int neutral_hip_set_arithmetic(int mode);
/* Resets particles 0..nparticles-1 of an existing store to their injected state
 * (same arguments as inject_particles, no allocation). */
void neutral_hip_reinject_particles(const int nparticles, const int local_nx,
                                    const int local_ny, const int pad,
                                    const double local_particle_left_off,
                                    const double local_particle_bottom_off,
                                    const double local_particle_width,
                                    const double local_particle_height,
                                    const int x_off, const int y_off,
                                    const double dt, const double* edgex,
                                    const double* edgey,
                                    const double initial_energy,
                                    NeutralHipParticle* particles);
/* Particle state and the tiled variant.  The tiled variant works on a private
 * array of records sorted by mesh tile that MIRRORS the SoA store of `particles`.
 * By default solve_transport_2d ends with a pass that writes the records back to the
 * SoA arrays (a permutation of 76 B per particle: 11 ms at 1e8 particles), so the
 * arrays are current whenever it returns, as the reference's are.  In the other
 * direction the arrays are read when the store is first stepped and after anything
 * the library can see rewriting them: inject/reinject, and writes through its own
 * hooks (copy_buffer, copy_int_buffer, neutral_hip_memcpy_h2d, neutral_hip_memset)
 * that land inside the store.  A caller that changes particle arrays BEHIND the
 * library's back (its own kernels, hipMemcpy) must say so with
 * neutral_hip_invalidate_particles() before the next solve_transport_2d; the next
 * step then re-imports the arrays (one pass over the store).
 * lazy != 0 drops that pass: the arrays are written back only by
 * neutral_hip_sync_particles() (or a call that needs them: another variant, reinject,
 * a read through the hooks, free): use it when nothing reads the arrays between
 * timesteps, as main.c with visit_dump = 0 (main.c:91-94,149-152).
 * Memory: the tally, density, edge and table arrays handed to solve_transport_2d must
 * be ordinary (coarse-grained) device memory, e.g. from hipMalloc or the hooks above:
 * the tally is accumulated with hardware f64 atomics (-munsafe-fp-atomics), which do
 * nothing on fine-grained or host-mapped memory. */
void neutral_hip_set_lazy_export(int lazy);
/* Tiled variant: on != 0 lets a history that leaves the tally window it streams under change
 * tiles INSIDE the stream kernel (a queue per tile, claimed by whichever workgroup is free:
 * "asynchronous tile queue") instead of waiting for another sort-and-stream pass.  Same
 * particle bits either way.  Off by default: measured level with the passes on the dense decks
 * and slower on the sparse ones.  Also: NEUTRAL_STREAM_QUEUES=1. */
void neutral_hip_set_stream_queues(int on);
void neutral_hip_sync_particles(NeutralHipParticle* particles);
void neutral_hip_invalidate_particles(NeutralHipParticle* particles);
/* ---- scalar-flux tally -------------------------------------------------------------
 * The reference declares `double* scalar_flux_tally` in NeutralData (neutral_data.h:95)
 * and never allocates or writes it, in any backend; the interface has no argument
 * for it.  This library keeps it when asked: the path-length estimator
 *     flux[cell] += (1 / ntotal_particles) * sum of weight * segment length
 * over the track segments a particle lays down in the cell (what collision, facet
 * and census events move it by; `weight` the weight it travels with), flushed to the
 * mesh exactly where the energy deposition is (facet, census, death).  Same layout
 * and normalisation as energy_deposition_tally: ny*nx doubles, accumulated, never
 * zeroed here, [device] coarse-grained memory.  NULL (default) turns it off, and
 * the kernels that run then are the ones without any flux code.  With it, the
 * tiled variant keeps two 88 x 88-cell windows in LDS instead of one of 128 x 128.
 * With several ranks it is all-reduced per step like the energy tally.
 * Checked against the CPU oracle's restatement of this definition and by what the
 * definition implies (tests/test_scalar_flux.py): in a collision-free deck
 * energy tally / flux is one constant, and the flux sums to speed * dt. */
void neutral_hip_set_scalar_flux_tally(double* device_tally);

/* ---- ranks: one process per GPU on one node ------------------------------------
 * The reference leaves rank and rank count to the parent project's initialise_mpi
 * (main.c:62) and calls barrier() (main.c:75,112) and reduce_all_sum
 * (omp3/neutral.c:530); its own MPI code is compiled out (neutral_data.h:10-14).
 * Here ranks are processes started by a launcher that exports RANK, WORLD_SIZE,
 * LOCAL_RANK, MASTER_ADDR, MASTER_PORT (torchrun's convention; `neutral.hip --gpus
 * N` forks them itself).  Particles are sharded over the ranks in contiguous id
 * ranges, the mesh is replicated, and solve_transport_2d ends every timestep with
 * ONE all-reduce of that step's tally contributions (sum, f64, nx*ny) over RCCL /
 * xGMI, after which energy_deposition_tally holds the same global mesh on every
 * rank -- all behind the unchanged three functions:
 *   inject_particles(nparticles = N, ...)  creates this rank's shard of the N
 *       particles (global ids first..first+count-1 as RNG keys);
 *   solve_transport_2d(...)                steps the shard it finds in `particles`
 *       (*nlocal_particles is not rewritten), all-reduces tally and event counters;
 *   validate(...)                          sums the (already global) tally.
 * The unchanged main.c gets there through the host layer linked into this library:
 * initialise_mpi reads the environment, initialise_devices binds the rank to GPU
 * LOCAL_RANK and calls neutral_hip_comm_start(). */
enum {
  NEUTRAL_HIP_COMM_NONE = 0, /* one rank */
  NEUTRAL_HIP_COMM_RCCL = 1, /* ncclAllReduce on the kernels' stream */
  NEUTRAL_HIP_COMM_HOST = 2  /* staged through the host over TCP (RCCL unavailable, or
                                NEUTRAL_HIP_COMM=host: ranks sharing a GPU in tests) */
};
/* Joins the ranks (TCP rendezvous at MASTER_ADDR:NEUTRAL_COMM_PORT, default
 * MASTER_PORT + 1), then brings up RCCL on the current device (select it first) with
 * a time limit (NEUTRAL_COMM_TIMEOUT seconds, default 120).  Returns the transport in
 * use (NEUTRAL_HIP_COMM_*).  Idempotent. */
int neutral_hip_comm_start(void);
void neutral_hip_comm_stop(void);
int neutral_hip_comm_rank(void);
int neutral_hip_comm_nranks(void);
int neutral_hip_comm_transport(void);
/* RCCL's version number as ncclGetVersion reports it (0: librccl is not loadable here) */
int neutral_hip_comm_rccl_version(void);
/* sharding by inject_particles when there are several ranks (default 1); 0 leaves the
 * id range to the caller (neutral_hip_set_pid_base + its own particle count) */
void neutral_hip_set_auto_shard(int on);
/* particles in a sharded or decomposed store created by inject_particles (this rank's
 * share, or what is inside its block right now); -1 for any other store */
int neutral_hip_store_count(const NeutralHipParticle* particles);
/* ---- spatial domain decomposition (mesh too large to replicate) -------------------
 * The reference has the plumbing only: rank offsets and neighbours in the interface
 * (neutral_interface.h:13-15), PARTICLE_SENT (neutral_data.h:35), a
 * send_and_mark_particle that is declared (omp3/neutral.h:63) and never defined; its
 * facet_event walks off the local arrays (omp3/neutral.c:333-377) and its RNG key is
 * the local array index (:89), so a decomposed run of it could not reproduce an
 * undecomposed one.  Here: ranks_x x ranks_y ranks (= the ranks of the rank layer) own
 * uniform blocks of the mesh; solve_transport_2d is called with the block's extent
 * (nx, ny, x_off, y_off, arrays of the block: edges nx+1 / ny+1, density and tally
 * nx*ny, pad = 0) and the global one (global_nx, global_ny).  A history that crosses
 * into another rank's block stops on the facet, is sent there with its RNG counter,
 * and goes on in the same timestep: rounds of exchange (RCCL send/recv; staged through
 * the host otherwise) until no rank has a history in flight.  Keys are global particle
 * ids, so every history is the one an undecomposed run computes, whichever ranks it
 * visits; tallies are per block (validate sums them over the ranks).  Tiled variant
 * only.
 *   neutral_hip_set_decomposition  names the grid (after neutral_hip_comm_start) and
 *       returns this rank's block; 0 on success, 1 if the grid does not match the ranks
 *   neutral_hip_set_source_box     the GLOBAL source box (same numbers on every rank);
 *   inject_particles(nparticles = N, ...) then makes a store with room for all N, holding
 *       the particles the source puts into this rank's block; their number changes as
 *       histories cross: solve_transport_2d writes it to *nlocal_particles, and
 *       neutral_hip_store_count / neutral_hip_store_keys give it and the ids ([device],
 *       keys[i] = id of the particle at index i of the arrays). */
int neutral_hip_set_decomposition(int ranks_x, int ranks_y, int global_nx, int global_ny,
                                  int* x_off, int* y_off, int* local_nx, int* local_ny);
void neutral_hip_clear_decomposition(void);
void neutral_hip_set_source_box(double left, double bottom, double width, double height);
const unsigned* neutral_hip_store_keys(const NeutralHipParticle* particles);

/* in-place sum over the ranks of n doubles in [device] memory, on hip_stream */
void neutral_hip_comm_allreduce_f64(double* device_buf, size_t n, void* hip_stream);
/* max over the ranks of a host scalar; barrier of the ranks (host side) */
double neutral_hip_comm_max(double v);
void neutral_hip_comm_barrier(void);
/* used by the host layer: bind this rank to its GPU and start the rank layer;
 * finish the device's work (the device half of barrier()) */
void neutral_hip_bind_rank_device(int local_rank);
void neutral_hip_comm_barrier_device(void);
/* one-rank RCCL check on the current device: 0 ok, 1 librccl not loadable, 2 failure */
int neutral_hip_comm_selftest(int n);

/* Frees a store created by inject_particles. */
void neutral_hip_free_particles(NeutralHipParticle* particles);
/* Raw copies for callers without a HIP runtime of their own (ctypes, C). */
void neutral_hip_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes);
void neutral_hip_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes);
void neutral_hip_memset(void* dst_device, int value, size_t bytes);
void neutral_hip_synchronize(void);
/* Unit probes of the device building blocks, for known-answer tests.  All
 * pointers are HOST arrays; the library stages them through HBM.
 *   threefry:  in3 = n rows {counter, pkey, master_key}; out2 = n rows of the
 *              two Threefry2x64-20 words; rn2 = the two (0,1] doubles of
 *              generate_random_numbers (omp3/neutral.c:632-652)
 *   cs_lookup: microscopic_cs_for_energy (omp3/neutral.c:498-517) of `cs`
 *              ([device] table) at n energies -> value and bracket index;
 *              use_index = 1 searches through the exponent-bucketed index the
 *              history kernels use, 0 by plain bisection
 *   distance_to_facet: in9 = n rows {x, y, omega_x, omega_y, speed, edgex[c],
 *              edgex[c+1], edgey[c], edgey[c+1]} (omp3/neutral.c:423-471) */
void neutral_hip_probe_threefry(const uint64_t* in3, uint64_t* out2, double* rn2, int n);
void neutral_hip_probe_cs_lookup(const NeutralHipCrossSection* cs, const double* energy,
                                 double* value, int* index, int n, int use_index);
void neutral_hip_probe_distance_to_facet(const double* in9, double* distance, int* x_facet,
                                         int n);
/*   division:  in2 = n rows {a, b}; out2 = n rows {a / b as the compiler divides,
 *              the quotient through the kept reciprocal of b (the stream kernel's
 *              form of omp3/neutral.c:311-312)}; plain[i] = 1 when both operands lie
 *              in the range where the kernel uses the second form */
void neutral_hip_probe_division(const double* in2, double* out2, int* plain, int n);
/*   log:       out8 = n rows {the logarithm the history kernels take of a sample
 *              (omp3/neutral.c:131,295), the device library's log, the kernels'
 *              square root (:255-259,297), the compiler's sqrt} of x[i], followed by
 *              n rows {x / PARTICLE_MASS the kernels' way (:297), as the compiler
 *              divides, x / (MASS_NO+1)^2 the kernels' way (:252), as the compiler
 *              divides} */
void neutral_hip_probe_log(const double* x, double* out8, int n);
/*   scatter:   in4 = n rows {energy, the centre-of-mass cosine mu, omega_x, omega_y}; out10 = n rows
 *              {the energy after the scatter (omp3/neutral.c:257-259), the laboratory cosine
 *              (:263-265) the fast kernels' way and with IEEE divisions and roots, the speed
 *              after the scatter (:297) from the speed before it and as sqrt(2 E' eV / m),
 *              1 / (omega_x speed) and 1 / (omega_y speed) (:435-436) off one reciprocal and as
 *              two divisions, 0}: what the fast arithmetic policy seeds from its neighbours */
void neutral_hip_probe_scatter(const double* in4, double* out10, int n);
/* Library/ABI version, bumped on any signature change. */
int neutral_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
