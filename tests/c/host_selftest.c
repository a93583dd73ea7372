/*
 * host_selftest.c -- exercises the plain-C host layer (deck reader, mesh,
 * density boxes, source box, cs reader, profiler) and the CPU oracle in one
 * process, so that the pair can be run under -fsanitize=address,undefined
 * (tests/test_sanitizers.py).  Not part of the product.
 *
 *   host_selftest <deck.params> <cs file> <tests file>
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../neutral_amd/host/comms.h"
#include "../../neutral_amd/host/mesh.h"
#include "../../neutral_amd/host/neutral_problem.h"
#include "../../neutral_amd/host/params.h"
#include "../../neutral_amd/host/profiler.h"
#include "../../neutral_amd/host/shared.h"
#include "../../neutral_amd/host/shared_data.h"
#include "../../oracle/neutral_oracle.h"

#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      fprintf(stderr, "selftest failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 4) {
    fprintf(stderr, "usage: host_selftest <deck> <cs file> <tests file>\n");
    return 2;
  }
  const char* deck = argv[1];

  Mesh mesh;
  memset(&mesh, 0, sizeof(mesh));
  mesh.global_nx = get_int_parameter("nx", deck);
  mesh.global_ny = get_int_parameter("ny", deck);
  mesh.local_nx = mesh.global_nx;
  mesh.local_ny = mesh.global_ny;
  mesh.width = 1.0;
  mesh.height = 1.0;
  mesh.dt = get_double_parameter("dt", deck);
  mesh.niters = get_int_parameter("iterations", deck);
  initialise_comms(&mesh);
  initialise_mesh_2d(&mesh);
  CHECK(mesh.edgex[0] == 0.0 && fabs(mesh.edgex[mesh.local_nx] - 1.0) < 1e-12);
  for (int ii = 0; ii < NNEIGHBOURS; ++ii) {
    CHECK(mesh.neighbours[ii] == EDGE);
  }

  SharedData shared = {0};
  initialise_shared_data_2d(mesh.local_nx, mesh.local_ny, 0, mesh.width, mesh.height, deck,
                            mesh.edgex, mesh.edgey, &shared);
  handle_boundary_2d(mesh.local_nx, mesh.local_ny, &mesh, shared.density, NO_INVERT, PACK);
  const int nx = mesh.local_nx, ny = mesh.local_ny;
  double dmin = 1e300, dmax = 0.0;
  for (int i = 0; i < nx * ny; ++i) {
    dmin = fmin(dmin, shared.density[i]);
    dmax = fmax(dmax, shared.density[i]);
  }
  CHECK(dmin > 0.0 && dmax >= dmin);

  NeutralSource src;
  neutral_source_from_deck(deck, mesh.width, mesh.height, mesh.edgex[0], mesh.edgey[0],
                           mesh.edgex[nx], mesh.edgey[ny], &src);
  CHECK(src.nlocal_particles == src.nparticles && src.nparticles > 0);

  /* cross-section file through the reader */
  const int nentries = neutral_cs_file_entries(argv[2]);
  CHECK(nentries == 29999);
  double* keys = (double*)malloc(sizeof(double) * nentries);
  double* values = (double*)malloc(sizeof(double) * nentries);
  CHECK(neutral_read_cs_file(argv[2], nentries, keys, values) == nentries);
  for (int i = 1; i < nentries; ++i) {
    CHECK(keys[i] > keys[i - 1]);
  }
  OrcCrossSection cs = {keys, values, nentries};
  CHECK(neutral_cs_file_entries("/nonexistent/file.cs") == -1);

  /* known-answer file lookup (omp3/neutral.c:541) */
  char* kv_keys = (char*)malloc(MAX_KEYS * MAX_STR_LEN);
  double kv_vals[MAX_KEYS];
  int nkv = 0;
  CHECK(get_key_value_parameter(deck, argv[3], kv_keys, kv_vals, &nkv) == 1 && nkv == 1);
  CHECK(get_key_value_parameter("no/such/deck", argv[3], kv_keys, kv_vals, &nkv) == 0);
  CHECK(within_tolerance(1.0, 1.0005, 1e-3) && !within_tolerance(1.0, 1.002, 1e-3));

  /* profiler: the global accumulating profile and the main.c-style stack profile */
  START_PROFILING(&compute_profile);
  STOP_PROFILING(&compute_profile, "selftest");
  START_PROFILING(&compute_profile);
  STOP_PROFILING(&compute_profile, "selftest");
  CHECK(compute_profile.profiler_entry_count == 1 && compute_profile.profiler_entries[0].calls == 2);
  struct Profile stack_profile; /* deliberately uninitialised, as main.c:82 */
  for (int tt = 1; tt <= 3; ++tt) {
    START_PROFILING(&stack_profile);
    const char p = '0' + tt;
    STOP_PROFILING(&stack_profile, &p);
    CHECK(stack_profile.profiler_entries[tt - 1].time >= 0.0);
  }

  /* the oracle on this problem: inject + all timesteps */
  const int n = src.nparticles;
  OrcParticles P;
  P.x = calloc(n, sizeof(double));
  P.y = calloc(n, sizeof(double));
  P.omega_x = calloc(n, sizeof(double));
  P.omega_y = calloc(n, sizeof(double));
  P.energy = calloc(n, sizeof(double));
  P.weight = calloc(n, sizeof(double));
  P.dt_to_census = calloc(n, sizeof(double));
  P.mfp_to_collision = calloc(n, sizeof(double));
  P.cellx = calloc(n, sizeof(int));
  P.celly = calloc(n, sizeof(int));
  P.dead = calloc(n, sizeof(int));
  double* tally = calloc((size_t)nx * ny, sizeof(double));
  orc_inject_particles(n, 0, nx, ny, 0, src.local_particle_left_off, src.local_particle_bottom_off,
                       src.local_particle_width, src.local_particle_height, 0, 0, mesh.dt,
                       mesh.edgex, mesh.edgey, src.initial_energy, &P);
  uint64_t facets = 0, collisions = 0, processed = 0;
  for (int tt = 1; tt <= mesh.niters; ++tt) {
    processed += orc_solve_transport_2d(nx, ny, nx, ny, (uint64_t)tt, 0, 0, 0, mesh.dt, n, n, 0,
                                        &P, shared.density, mesh.edgex, mesh.edgey, &cs, &cs,
                                        tally, &facets, &collisions);
  }
  for (int i = 0; i < n; ++i) {
    CHECK(P.cellx[i] >= 0 && P.cellx[i] < nx && P.celly[i] >= 0 && P.celly[i] < ny);
    CHECK(P.x[i] >= -1e-9 && P.x[i] <= 1.0 + 1e-9);
  }
  const double total = orc_sum_tally(nx, ny, tally);
  CHECK(total > 0.0 && isfinite(total));
  printf("selftest ok: processed %llu facets %llu collisions %llu tally %.15e\n",
         (unsigned long long)processed, (unsigned long long)facets,
         (unsigned long long)collisions, total);

  free(tally);
  free(P.x); free(P.y); free(P.omega_x); free(P.omega_y); free(P.energy); free(P.weight);
  free(P.dt_to_census); free(P.mfp_to_collision); free(P.cellx); free(P.celly); free(P.dead);
  free(kv_keys);
  free(keys);
  free(values);
  deallocate_data(shared.density);
  deallocate_data(shared.energy);
  deallocate_data(mesh.edgex); deallocate_data(mesh.edgey);
  deallocate_data(mesh.edgedx); deallocate_data(mesh.edgedy);
  deallocate_data(mesh.celldx); deallocate_data(mesh.celldy);
  return 0;
}
