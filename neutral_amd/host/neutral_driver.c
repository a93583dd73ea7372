/*
 * neutral_driver.c -- `neutral.hip`: stand-alone driver of the MI355X kernel
 * set, for machines where the reference tree is not at hand.  It drives the
 * three interface functions (neutral_interface.h:11-36) the way the reference's
 * main.c + neutral_data.c do -- deck, mesh, density, source box, injection,
 * cross-section tables, timestep loop, validation -- and prints the same
 * per-step lines (main.c:88,118-125,161-162), so logs are comparable line by
 * line.  Everything numerical happens behind the C ABI (include/neutral_hip.h).
 *
 *   neutral.hip <deck.params> [--set key=value ...] [--arch-params FILE]
 *               [--cs-dir DIR] [--tests FILE] [--variant 0|1|2] [--gpus N]
 *               [--decompose PXxPY]
 *
 * --gpus N runs N ranks, one per GPU of this node: the driver forks them before
 * anything touches a GPU (ranks are ordinary processes that find each other through
 * RANK / WORLD_SIZE / MASTER_PORT, so any launcher that exports those -- torchrun
 * --no-python, for one -- does as well); particles are sharded, every rank holds
 * the mesh, and each timestep ends with one all-reduce of the tally (RCCL).
 * --decompose PXxPY (PX * PY = N) cuts the MESH over the ranks instead: every rank
 * holds one block of it and the particles inside; histories that cross between
 * blocks are exchanged within the timestep (include/neutral_hip.h).
 *
 * --set overrides a scalar deck entry (nx, ny, nparticles, iterations, dt,
 * initial_energy): the BASELINE configurations are the shipped decks at other
 * sizes.  ../arch.params (neutral_data.h:32) supplies width/height/sim_end when
 * present; otherwise 1.0 x 1.0, the extent the reference's known answers need.
 */
#include <arpa/inet.h>
#include <netinet/in.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "../../include/neutral_hip.h"
#include "comms.h"
#include "mesh.h"
#include "neutral_problem.h"
#include "params.h"
#include "shared.h"
#include "shared_data.h"

#define MAX_OVERRIDES 16

static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1.0e-9 * (double)ts.tv_nsec;
}

/* Writes a copy of `deck` with the scalar entries named in keys[] replaced. */
static void write_patched_deck(const char* deck, const char* out, int n, char keys[][64],
                               char values[][64]) {
  FILE* in = fopen(deck, "r");
  if (!in) {
    TERMINATE("Could not open the parameter file: %s.\n", deck);
  }
  FILE* fp = fopen(out, "w");
  if (!fp) {
    TERMINATE("Could not write %s.\n", out);
  }
  char line[4096];
  int used[MAX_OVERRIDES] = {0};
  while (fgets(line, sizeof(line), in)) {
    char first[256] = "";
    sscanf(line, " %255s", first);
    int replaced = 0;
    for (int k = 0; k < n; ++k) {
      if (strcmp(first, keys[k]) == 0) {
        fprintf(fp, "%s %s\n", keys[k], values[k]);
        used[k] = 1;
        replaced = 1;
      }
    }
    if (!replaced) {
      fputs(line, fp);
    }
  }
  for (int k = 0; k < n; ++k) {
    if (!used[k]) {
      fprintf(fp, "%s %s\n", keys[k], values[k]);
    }
  }
  fclose(in);
  fclose(fp);
}

/* a TCP port nobody listens on right now (for the ranks' rendezvous) */
static int free_port(void) {
  const int fd = socket(AF_INET, SOCK_STREAM, 0);
  struct sockaddr_in sa;
  memset(&sa, 0, sizeof(sa));
  sa.sin_family = AF_INET;
  sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  socklen_t len = sizeof(sa);
  int port = 29611;
  if (fd >= 0 && bind(fd, (struct sockaddr*)&sa, sizeof(sa)) == 0 &&
      getsockname(fd, (struct sockaddr*)&sa, &len) == 0) {
    port = ntohs(sa.sin_port);
  }
  if (fd >= 0) close(fd);
  return port;
}

/* --gpus N: becomes rank 0..N-1 in N child processes (returns in the children with
 * the rank's environment set); the parent waits for them and exits with the worst
 * status.  Called before anything initialises a GPU. */
static void fork_ranks(int nranks) {
  char buf[32];
  snprintf(buf, sizeof(buf), "%d", free_port());
  setenv("MASTER_ADDR", "127.0.0.1", 1);
  setenv("MASTER_PORT", buf, 1);
  /* (the port that was just found free is the one the ranks meet on: left to its default --
   * MASTER_PORT + 1, comms_ranks.c -- the rendezvous would listen on a port nobody has looked at) */
  setenv("NEUTRAL_COMM_PORT", buf, 1);
  snprintf(buf, sizeof(buf), "%d", nranks);
  setenv("WORLD_SIZE", buf, 1);
  if (!getenv("NEUTRAL_COMM_NONCE")) {
    /* the word the ranks of THIS launch greet rank 0 with (comms_ranks.c) */
    unsigned long long nonce = ((unsigned long long)getpid() << 32) ^ (unsigned long long)time(NULL);
    FILE* rnd = fopen("/dev/urandom", "rb");
    if (rnd) {
      if (fread(&nonce, sizeof(nonce), 1, rnd) != 1) { /* keep the fallback */ }
      fclose(rnd);
    }
    snprintf(buf, sizeof(buf), "%llu", nonce);
    setenv("NEUTRAL_COMM_NONCE", buf, 1);
  }
  pid_t kids[64];
  for (int r = 0; r < nranks; ++r) {
    fflush(stdout);
    const pid_t pid = fork();
    if (pid < 0) {
      TERMINATE("Could not start rank %d.\n", r);
    }
    if (pid == 0) {
      snprintf(buf, sizeof(buf), "%d", r);
      setenv("RANK", buf, 1);
      setenv("LOCAL_RANK", buf, 1);
      return;
    }
    kids[r] = pid;
  }
  int worst = 0;
  for (int r = 0; r < nranks; ++r) {
    int status = 0;
    waitpid(kids[r], &status, 0);
    const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + WTERMSIG(status);
    worst = (code > worst) ? code : worst;
  }
  exit(worst);
}

static void load_table(const char* path, NeutralHipCrossSection* cs) {
  const int n = neutral_cs_file_entries(path);
  if (n < 0) {
    TERMINATE("Could not open the cross section file: %s\n", path);
  }
  if (comms_rank() == MASTER) {
    printf("File %s contains %d entries\n", path, n); /* neutral_data.c:139 */
  }
  double* h_keys;
  double* h_values;
  allocate_host_data(&h_keys, (size_t)n);
  allocate_host_data(&h_values, (size_t)n);
  cs->nentries = neutral_read_cs_file(path, n, h_keys, h_values);
  move_host_buffer_to_device((size_t)cs->nentries, &h_keys, &cs->keys);
  move_host_buffer_to_device((size_t)cs->nentries, &h_values, &cs->values);
}

int main(int argc, char** argv) {
  if (argc < 2) {
    TERMINATE("usage: ./neutral.hip <param_file> [--set key=value ...] [--arch-params FILE] "
              "[--cs-dir DIR] [--tests FILE] [--variant N]\n");
  }
  const char* deck = argv[1];
  const char* arch_params = "../arch.params";
  const char* cs_dir = ".";
  char keys[MAX_OVERRIDES][64];
  char values[MAX_OVERRIDES][64];
  int noverrides = 0;
  int decompose_x = 0, decompose_y = 0;
  /* multi-process GPU work on this stack needs dmabuf IPC; read by the runtime at start-up */
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
  for (int i = 2; i + 1 < argc; ++i) {
    if (strcmp(argv[i], "--gpus") == 0) {
      const int n = atoi(argv[i + 1]);
      if (n < 1 || n > 64) {
        TERMINATE("--gpus wants 1..64\n");
      }
      if (n > 1 && !getenv("RANK")) {
        fork_ranks(n);
      }
    }
  }
  for (int i = 2; i < argc; ++i) {
    if (strcmp(argv[i], "--set") == 0 && i + 1 < argc && noverrides < MAX_OVERRIDES) {
      char* eq = strchr(argv[++i], '=');
      if (!eq) {
        TERMINATE("--set needs key=value\n");
      }
      snprintf(keys[noverrides], sizeof(keys[0]), "%.*s", (int)(eq - argv[i]), argv[i]);
      snprintf(values[noverrides], sizeof(values[0]), "%s", eq + 1);
      noverrides++;
    } else if (strcmp(argv[i], "--arch-params") == 0 && i + 1 < argc) {
      arch_params = argv[++i];
    } else if (strcmp(argv[i], "--cs-dir") == 0 && i + 1 < argc) {
      cs_dir = argv[++i];
    } else if (strcmp(argv[i], "--tests") == 0 && i + 1 < argc) {
      neutral_hip_set_tests_file(argv[++i]);
    } else if (strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) {
      ++i; /* handled above, before anything touched a GPU */
    } else if (strcmp(argv[i], "--decompose") == 0 && i + 1 < argc) {
      if (sscanf(argv[++i], "%dx%d", &decompose_x, &decompose_y) != 2) {
        TERMINATE("--decompose wants PXxPY, e.g. 4x2\n");
      }
    } else if (strcmp(argv[i], "--variant") == 0 && i + 1 < argc) {
      if (neutral_hip_set_variant(atoi(argv[++i]))) {
        TERMINATE("unknown --variant\n");
      }
    } else {
      TERMINATE("unknown argument %s\n", argv[i]);
    }
  }

  /* deck actually read: the original, or a patched copy (one per rank) */
  char patched[4096];
  const char* read_deck = deck;
  if (noverrides) {
    snprintf(patched, sizeof(patched), "/tmp/neutral_hip_deck_%ld.params", (long)getpid());
    write_patched_deck(deck, patched, noverrides, keys, values);
    read_deck = patched;
  }

  Mesh mesh;
  memset(&mesh, 0, sizeof(mesh));
  mesh.global_nx = get_int_parameter("nx", read_deck);
  mesh.global_ny = get_int_parameter("ny", read_deck);
  mesh.pad = 0;
  mesh.local_nx = mesh.global_nx + 2 * mesh.pad;
  mesh.local_ny = mesh.global_ny + 2 * mesh.pad;
  mesh.width = 1.0;
  mesh.height = 1.0;
  mesh.sim_end = 1.0e30;
  (void)try_get_double_parameter("width", arch_params, &mesh.width);
  (void)try_get_double_parameter("height", arch_params, &mesh.height);
  (void)try_get_double_parameter("sim_end", arch_params, &mesh.sim_end);
  mesh.dt = get_double_parameter("dt", read_deck);
  mesh.niters = get_int_parameter("iterations", read_deck);
  mesh.rank = MASTER;
  mesh.nranks = 1;
  mesh.ndims = 2;

  initialise_mpi(argc, argv, &mesh.rank, &mesh.nranks);
  const int master = (mesh.rank == MASTER);
  if (master) {
    printf("Starting up with %d rank(s), one per GPU, kernel set: libneutral_hip (gfx950).\n",
           mesh.nranks);
    printf("Loading problem from %s.\n", deck);
  } else {
    neutral_hip_set_quiet(1); /* one "Particles" line per step: rank 0's */
  }
  initialise_devices(mesh.rank); /* binds the rank to its GPU, starts the tally exchange */
  initialise_comms(&mesh);
  if (decompose_x) {
    /* this rank's block of the mesh instead of all of it */
    int xo, yo, lx, ly;
    if (neutral_hip_set_decomposition(decompose_x, decompose_y, mesh.global_nx, mesh.global_ny,
                                      &xo, &yo, &lx, &ly)) {
      TERMINATE("--decompose %dx%d does not fit %d rank(s) and a %d x %d mesh.\n", decompose_x,
                decompose_y, mesh.nranks, mesh.global_nx, mesh.global_ny);
    }
    mesh.x_off = xo;
    mesh.y_off = yo;
    mesh.local_nx = lx + 2 * mesh.pad;
    mesh.local_ny = ly + 2 * mesh.pad;
  }
  initialise_mesh_2d(&mesh);
  SharedData shared_data = {0};
  initialise_shared_data_2d(mesh.local_nx, mesh.local_ny, mesh.pad, mesh.width, mesh.height,
                            read_deck, mesh.edgex, mesh.edgey, &shared_data);
  handle_boundary_2d(mesh.local_nx, mesh.local_ny, &mesh, shared_data.density, NO_INVERT, PACK);

  /* source box and particle count: four edge scalars come back from HBM */
  const int nx = mesh.local_nx - 2 * mesh.pad;
  const int ny = mesh.local_ny - 2 * mesh.pad;
  double edges[4];
  double* h = NULL;
  allocate_host_data(&h, 1);
  /* (the edge arrays hold this rank's edges: local indices) */
  double* d_edge[4] = {&mesh.edgex[mesh.pad], &mesh.edgey[mesh.pad], &mesh.edgex[nx + mesh.pad],
                       &mesh.edgey[ny + mesh.pad]};
  for (int k = 0; k < 4; ++k) {
    copy_buffer(1, &d_edge[k], &h, RECV);
    edges[k] = *h;
  }
  deallocate_host_data(h);
  if (decompose_x) {
    /* the library makes every rank look at all source particles and keep those of its
     * block: it wants the source box of the whole mesh */
    edges[0] = 0.0;
    edges[1] = 0.0;
    edges[2] = mesh.width;
    edges[3] = mesh.height;
  }
  NeutralSource src;
  neutral_source_from_deck(read_deck, mesh.width, mesh.height, edges[0], edges[1], edges[2],
                           edges[3], &src);

  if (decompose_x) {
    neutral_hip_set_source_box(src.local_particle_left_off, src.local_particle_bottom_off,
                               src.local_particle_width, src.local_particle_height);
  }
  double* tally = NULL;
  size_t allocation = allocate_data(&tally, (size_t)nx * (size_t)ny);
  NeutralHipParticle* particles = NULL;
  int nlocal = src.nlocal_particles;
  if (nlocal) {
    allocation += inject_particles(src.nparticles, mesh.global_nx, mesh.local_nx, mesh.local_ny,
                                   mesh.pad, src.local_particle_left_off,
                                   src.local_particle_bottom_off, src.local_particle_width,
                                   src.local_particle_height, mesh.x_off, mesh.y_off, mesh.dt,
                                   mesh.edgex, mesh.edgey, src.initial_energy, &particles);
    if (decompose_x) {
      nlocal = neutral_hip_store_count(particles); /* what the source put into this block */
    }
  }
  if (master) {
    printf("Allocated %.4fGB of data.\n", allocation / GB); /* neutral_data.c:117 */
  }

  NeutralHipCrossSection cs_scatter, cs_absorb;
  char path[4096];
  snprintf(path, sizeof(path), "%s/elastic_scatter.cs", cs_dir); /* neutral_data.h:30 */
  load_table(path, &cs_scatter);
  snprintf(path, sizeof(path), "%s/capture.cs", cs_dir); /* neutral_data.h:31 */
  load_table(path, &cs_absorb);

  /* timestep loop, main.c:85-147 */
  neutral_hip_set_lazy_export(1); /* nothing reads the particle arrays between steps */
  double wallclock = 0.0;
  double elapsed_sim_time = 0.0;
  int tt;
  for (tt = 1; tt <= mesh.niters; ++tt) {
    if (master) {
      printf("\nIteration  %d\n", tt); /* main.c:87-89 */
    }
    uint64_t facet_events = 0;
    uint64_t collision_events = 0;
    const double t0 = now_seconds();
    solve_transport_2d(nx, ny, mesh.global_nx, mesh.global_ny, (uint64_t)tt, mesh.pad,
                       mesh.x_off, mesh.y_off, mesh.dt, src.nparticles, &nlocal,
                       mesh.neighbours, particles, shared_data.density, mesh.edgex, mesh.edgey,
                       mesh.edgedx, mesh.edgedy, &cs_scatter, &cs_absorb, tally, NULL, NULL,
                       NULL, &facet_events, &collision_events);
    barrier();
    const double step_time = now_seconds() - t0;
    wallclock += step_time;
    if (master) {
      /* (event counts are the sums over all ranks) */
      printf("Step time  %.4fs\n", step_time);
      printf("Wallclock  %.4fs\n", wallclock);
      printf("Facets     %llu\n", (unsigned long long)facet_events);
      printf("Collisions %llu\n", (unsigned long long)collision_events);
      printf("Facet Events / s %.2e\n", facet_events / step_time);
      printf("Collision Events / s %.2e\n", collision_events / step_time);
      NeutralHipStepStats st;
      neutral_hip_last_step(&st);
      printf("Particle-steps / s %.3e (facets + collisions + census, kernels %.2f ms)\n",
             (double)(st.facets + st.collisions + st.census) / step_time, st.kernel_ms);
    }
    elapsed_sim_time += mesh.dt;
    if (elapsed_sim_time >= mesh.sim_end) {
      if (master) {
        printf("Reached end of simulation time\n");
      }
      break;
    }
  }

  neutral_hip_sync_particles(particles);
  validate(nx, ny, deck, mesh.rank, tally);
  if (master) {
    printf("Final Wallclock %.9fs\n", wallclock);
    printf("Elapsed Simulation Time %.6fs\n", elapsed_sim_time);
  }
  if (noverrides) {
    remove(patched);
  }
  barrier();
  neutral_hip_comm_stop();
  return 0;
}
