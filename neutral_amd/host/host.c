/*
 * host.c -- implementation of the arch-compatible host layer (see shared.h).
 *
 * None of this is in the reference tree; what each routine must do is taken
 * from its call sites in main.c / neutral_data.c / omp3/neutral.c, and the
 * three free choices (mesh extent, edge formula, problem-box rule) are pinned
 * by the reference's own known answers (problems/neutral.tests; SURVEY.md
 * sections 0.3 and 8(b)).
 */
#include "comms.h"
#include "mesh.h"
#include "params.h"
#include "profiler.h"
#include "shared.h"
#include "shared_data.h"

#include <ctype.h>
#include <math.h>
#include <string.h>
#include <time.h>

/* ---- shared.h -------------------------------------------------------------- */

int within_tolerance(const double expected, const double result,
                     const double tolerance) {
  /* relative difference; an exact zero expectation falls back to absolute */
  if (expected == 0.0) {
    return fabs(result) < tolerance;
  }
  return fabs((result - expected) / expected) < tolerance;
}

/* ---- profiler.h ------------------------------------------------------------ */

struct Profile compute_profile = {0};

static double wall_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1.0e-9 * (double)ts.tv_nsec;
}

void profiler_start_timer(struct Profile* profile) {
  profile->start_seconds = wall_seconds();
}

void profiler_end_timer(struct Profile* profile, const char* entry_name) {
  const double elapsed = wall_seconds() - profile->start_seconds;

  if (profile != &compute_profile) {
    /* main.c:82 declares its `struct Profile` on the stack without
     * initialising it, names each step with ONE unterminated character
     * '0'+tt (main.c:114-115) and then reads profiler_entries[tt-1].time
     * (main.c:116).  The only behaviour that is correct for that caller is
     * to index by that character and to SET the slot. */
    const unsigned idx = (unsigned char)(entry_name[0] - '1');
    if (idx < PROFILER_MAX_ENTRIES) {
      profile->profiler_entries[idx].time = elapsed;
      profile->profiler_entries[idx].calls = 1;
      profile->profiler_entries[idx].name[0] = entry_name[0];
      profile->profiler_entries[idx].name[1] = '\0';
    }
    return;
  }

  /* named, accumulating entries for the zero-initialised global profile */
  int ii;
  for (ii = 0; ii < profile->profiler_entry_count; ++ii) {
    if (strncmp(profile->profiler_entries[ii].name, entry_name,
                PROFILER_MAX_NAME - 1) == 0) {
      break;
    }
  }
  if (ii == profile->profiler_entry_count) {
    if (ii == PROFILER_MAX_ENTRIES) {
      return;
    }
    profile->profiler_entry_count++;
    strncpy(profile->profiler_entries[ii].name, entry_name,
            PROFILER_MAX_NAME - 1);
    profile->profiler_entries[ii].name[PROFILER_MAX_NAME - 1] = '\0';
    profile->profiler_entries[ii].time = 0.0;
    profile->profiler_entries[ii].calls = 0;
  }
  profile->profiler_entries[ii].time += elapsed;
  profile->profiler_entries[ii].calls++;
}

void profiler_print_full_profile(struct Profile* profile) {
  printf("\n%-40s%12s%8s\n", "Profile entry", "time (s)", "calls");
  for (int ii = 0; ii < profile->profiler_entry_count; ++ii) {
    printf("%-40s%12.4f%8d\n", profile->profiler_entries[ii].name,
           profile->profiler_entries[ii].time,
           profile->profiler_entries[ii].calls);
  }
}

/* ---- params.h -------------------------------------------------------------- */

/* Copies the first whitespace-delimited token of `line` into tok (size n) and
 * returns a pointer just past it, or NULL for an empty / comment line. */
static const char* first_token(const char* line, char* tok, size_t n) {
  while (*line == ' ' || *line == '\t') {
    line++;
  }
  if (*line == '\0' || *line == '\n' || *line == '\r' || *line == '#') {
    return NULL;
  }
  size_t len = 0;
  while (*line && !isspace((unsigned char)*line) && *line != '#') {
    if (len + 1 < n) {
      tok[len++] = *line;
    }
    line++;
  }
  tok[len] = '\0';
  return line;
}

/* Finds the entry line for `name`; on success copies the remainder of the
 * line (comment stripped) into rest. */
static int find_entry(const char* name, const char* filename, char* rest,
                      size_t n) {
  FILE* fp = fopen(filename, "r");
  if (!fp) {
    return -1;
  }
  char line[4 * MAX_STR_LEN];
  char tok[MAX_STR_LEN];
  int found = 0;
  while (fgets(line, sizeof(line), fp)) {
    const char* after = first_token(line, tok, sizeof(tok));
    if (!after || strcmp(tok, name) != 0) {
      continue;
    }
    size_t len = 0;
    while (*after && *after != '#' && *after != '\n' && *after != '\r') {
      if (len + 1 < n) {
        rest[len++] = *after;
      }
      after++;
    }
    rest[len] = '\0';
    found = 1;
    break;
  }
  fclose(fp);
  return found;
}

int try_get_double_parameter(const char* param_name, const char* filename,
                             double* value) {
  char rest[4 * MAX_STR_LEN];
  if (find_entry(param_name, filename, rest, sizeof(rest)) != 1) {
    return 0;
  }
  char* end = NULL;
  const double v = strtod(rest, &end);
  if (end == rest) {
    return 0;
  }
  *value = v;
  return 1;
}

double get_double_parameter(const char* param_name, const char* filename) {
  char rest[4 * MAX_STR_LEN];
  const int status = find_entry(param_name, filename, rest, sizeof(rest));
  if (status < 0) {
    TERMINATE("Could not open the parameter file: %s.\n", filename);
  }
  if (status == 0) {
    TERMINATE("Parameter %s was not found in %s.\n", param_name, filename);
  }
  char* end = NULL;
  const double v = strtod(rest, &end);
  if (end == rest) {
    TERMINATE("Parameter %s in %s has no numeric value.\n", param_name,
              filename);
  }
  return v;
}

int get_int_parameter(const char* param_name, const char* filename) {
  const double v = get_double_parameter(param_name, filename);
  return (int)llround(v);
}

int get_key_value_parameter(const char* specifier, const char* filename,
                            char* keys, double* values, int* nkeys) {
  char rest[4 * MAX_STR_LEN];
  *nkeys = 0;
  if (find_entry(specifier, filename, rest, sizeof(rest)) != 1) {
    return 0;
  }
  char* save = NULL;
  for (char* tok = strtok_r(rest, " \t", &save); tok && *nkeys < MAX_KEYS;
       tok = strtok_r(NULL, " \t", &save)) {
    char* eq = strchr(tok, '=');
    if (!eq) {
      continue;
    }
    *eq = '\0';
    strncpy(&keys[(*nkeys) * MAX_STR_LEN], tok, MAX_STR_LEN - 1);
    keys[(*nkeys) * MAX_STR_LEN + MAX_STR_LEN - 1] = '\0';
    values[*nkeys] = strtod(eq + 1, NULL);
    (*nkeys)++;
  }
  return 1;
}

/* ---- comms.h --------------------------------------------------------------- */

void initialise_mpi(int argc, char** argv, int* rank, int* nranks) {
  (void)argc;
  (void)argv;
  comms_start_from_env();
  *rank = comms_rank();
  *nranks = comms_nranks();
}

/* binds the rank to its GPU when a device kernel set is linked (weak: the
 * host-memory flavour of the library has no devices) */
void neutral_hip_bind_rank_device(int local_rank) __attribute__((weak));
void neutral_hip_comm_barrier_device(void) __attribute__((weak));

void initialise_devices(int rank) {
  (void)rank;
  if (neutral_hip_bind_rank_device) {
    neutral_hip_bind_rank_device(comms_local_rank());
  }
}

void initialise_comms(Mesh* mesh) {
  /* every rank owns the whole mesh (particles are sharded, not cells): no offsets,
   * every neighbour is the edge */
  for (int ii = 0; ii < NNEIGHBOURS; ++ii) {
    mesh->neighbours[ii] = EDGE;
  }
  mesh->x_off = 0;
  mesh->y_off = 0;
}

void finalise_comms(void) { comms_stop(); }

void barrier(void) {
  /* main.c:75,112: "everybody has finished the step"; the device's work is part of it */
  if (neutral_hip_comm_barrier_device) {
    neutral_hip_comm_barrier_device();
  }
  comms_barrier();
}

double reduce_all_sum(double local_val) {
  comms_allreduce_f64(&local_val, 1, COMMS_SUM);
  return local_val;
}
double reduce_all_min(double local_val) {
  comms_allreduce_f64(&local_val, 1, COMMS_MIN);
  return local_val;
}
double reduce_all_max(double local_val) {
  comms_allreduce_f64(&local_val, 1, COMMS_MAX);
  return local_val;
}

void handle_boundary_2d(const int nx, const int ny, Mesh* mesh, double* arr,
                        const int invert, const int pack) {
  /* halo exchange / reflective fill of `pad` ghost layers.  neutral runs with
   * pad = 0 (main.c:33), for which there is nothing to do. */
  (void)nx;
  (void)ny;
  (void)arr;
  (void)invert;
  (void)pack;
  if (mesh->pad != 0) {
    TERMINATE("handle_boundary_2d: pad != 0 is not supported by this host "
              "layer (neutral uses pad = 0).\n");
  }
}

void write_all_ranks_to_visit(const int global_nx, const int global_ny,
                              const int local_nx, const int local_ny,
                              const int pad, const int x_off, const int y_off,
                              const int rank, const int nranks,
                              int* neighbours, double* local_arr,
                              const char* name, const int tt,
                              const double elapsed_sim_time) {
  (void)global_nx;
  (void)global_ny;
  (void)local_nx;
  (void)local_ny;
  (void)pad;
  (void)x_off;
  (void)y_off;
  (void)nranks;
  (void)neighbours;
  (void)local_arr;
  (void)tt;
  (void)elapsed_sim_time;
  if (rank == MASTER) {
    printf("visit_dump of '%s' skipped: VisIt output is not part of this host "
           "layer.\n",
           name);
  }
}

/* ---- mesh.h ---------------------------------------------------------------- */

void initialise_mesh_2d(Mesh* mesh) {
  const int nxp1 = mesh->local_nx + 1;
  const int nyp1 = mesh->local_ny + 1;

  double* h_edgex = (double*)malloc(sizeof(double) * nxp1);
  double* h_edgey = (double*)malloc(sizeof(double) * nyp1);
  double* h_edgedx = (double*)malloc(sizeof(double) * nxp1);
  double* h_edgedy = (double*)malloc(sizeof(double) * nyp1);
  double* h_celldx = (double*)malloc(sizeof(double) * nxp1);
  double* h_celldy = (double*)malloc(sizeof(double) * nyp1);
  if (!h_edgex || !h_edgey || !h_edgedx || !h_edgedy || !h_celldx ||
      !h_celldy) {
    TERMINATE("Could not allocate the mesh edges.\n");
  }

  /* uniform spacing; edge i of the local (padded) array sits at global
   * index x_off + i - pad */
  for (int ii = 0; ii < nxp1; ++ii) {
    h_edgedx[ii] = mesh->width / (double)mesh->global_nx;
    h_celldx[ii] = mesh->width / (double)mesh->global_nx;
    h_edgex[ii] = h_edgedx[ii] * (double)(mesh->x_off + ii - mesh->pad);
  }
  for (int ii = 0; ii < nyp1; ++ii) {
    h_edgedy[ii] = mesh->height / (double)mesh->global_ny;
    h_celldy[ii] = mesh->height / (double)mesh->global_ny;
    h_edgey[ii] = h_edgedy[ii] * (double)(mesh->y_off + ii - mesh->pad);
  }

  move_host_buffer_to_device(nxp1, &h_edgex, &mesh->edgex);
  move_host_buffer_to_device(nyp1, &h_edgey, &mesh->edgey);
  move_host_buffer_to_device(nxp1, &h_edgedx, &mesh->edgedx);
  move_host_buffer_to_device(nyp1, &h_edgedy, &mesh->edgedy);
  move_host_buffer_to_device(nxp1, &h_celldx, &mesh->celldx);
  move_host_buffer_to_device(nyp1, &h_celldy, &mesh->celldy);
}

/* ---- shared_data.h --------------------------------------------------------- */

void initialise_shared_data_2d(const int local_nx, const int local_ny,
                               const int pad, const double mesh_width,
                               const double mesh_height,
                               const char* problem_def_filename,
                               const double* edgex, const double* edgey,
                               SharedData* shared_data) {
  (void)pad;
  const size_t ncells = (size_t)local_nx * (size_t)local_ny;

  /* edges may live in device memory: fetch host copies through the hook */
  double* h_edgex = NULL;
  double* h_edgey = NULL;
  allocate_host_data(&h_edgex, (size_t)local_nx + 1);
  allocate_host_data(&h_edgey, (size_t)local_ny + 1);
  double* d_edgex = (double*)edgex;
  double* d_edgey = (double*)edgey;
  copy_buffer((size_t)local_nx + 1, &d_edgex, &h_edgex, RECV);
  copy_buffer((size_t)local_ny + 1, &d_edgey, &h_edgey, RECV);

  double* h_density = (double*)calloc(ncells, sizeof(double));
  double* h_energy = (double*)calloc(ncells, sizeof(double));
  if (!h_density || !h_energy) {
    TERMINATE("Could not allocate the shared data.\n");
  }

  char* keys = (char*)malloc(sizeof(char) * MAX_KEYS * MAX_STR_LEN);
  double* values = (double*)malloc(sizeof(double) * MAX_KEYS);
  if (!keys || !values) {
    TERMINATE("Could not allocate the problem entry buffers.\n");
  }

  /* problem_0, problem_1, ... applied in order; a later box overrides an
   * earlier one.  As for the deck's `source` entry (neutral_data.c:39-43) the
   * LAST four values are the box x, y, width, height as fractions of the mesh
   * extent; density/energy are looked up by key name.  A cell belongs to a box
   * when its lower-left corner does. */
  for (int pp = 0;; ++pp) {
    char specifier[64];
    snprintf(specifier, sizeof(specifier), "problem_%d", pp);
    int nkeys = 0;
    if (!get_key_value_parameter(specifier, problem_def_filename, keys, values,
                                 &nkeys)) {
      if (pp == 0) {
        TERMINATE("Parameter file %s did not contain a problem_0 entry.\n",
                  problem_def_filename);
      }
      break;
    }
    if (nkeys < 4) {
      TERMINATE("Entry %s of %s needs xpos, ypos, width and height.\n",
                specifier, problem_def_filename);
    }

    const double xpos = values[nkeys - 4] * mesh_width;
    const double ypos = values[nkeys - 3] * mesh_height;
    const double width = values[nkeys - 2] * mesh_width;
    const double height = values[nkeys - 1] * mesh_height;

    int has_density = 0, has_energy = 0;
    double density = 0.0, energy = 0.0;
    for (int kk = 0; kk < nkeys; ++kk) {
      const char* key = &keys[kk * MAX_STR_LEN];
      if (strcmp(key, "density") == 0) {
        density = values[kk];
        has_density = 1;
      } else if (strcmp(key, "energy") == 0) {
        energy = values[kk];
        has_energy = 1;
      }
    }

    for (int ii = 0; ii < local_ny; ++ii) {
      for (int jj = 0; jj < local_nx; ++jj) {
        const double cx = h_edgex[jj];
        const double cy = h_edgey[ii];
        if (cx >= xpos && cx < xpos + width && cy >= ypos &&
            cy < ypos + height) {
          if (has_density) {
            h_density[(size_t)ii * local_nx + jj] = density;
          }
          if (has_energy) {
            h_energy[(size_t)ii * local_nx + jj] = energy;
          }
        }
      }
    }
  }

  free(keys);
  free(values);
  deallocate_host_data(h_edgex);
  deallocate_host_data(h_edgey);

  move_host_buffer_to_device(ncells, &h_density, &shared_data->density);
  move_host_buffer_to_device(ncells, &h_energy, &shared_data->energy);
}
