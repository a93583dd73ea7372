/*
 * neutral_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see the header).
 *
 * CPU restatement of UoB-HPC/neutral omp3/neutral.c.  Floating-point
 * expressions keep the reference's association order so that the compiler
 * sees the same operation sequence; histories are kept in locals and written
 * back once, which is unobservable because nothing else reads a particle
 * while its history runs (omp3/neutral.c:78-198 touches only particle `pid`).
 */
#include "neutral_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>

/* ---- Threefry2x64-20 ---------------------------------------------------- */

/* Random123/threefry.h:158-162 */
static inline uint64_t rotl64(uint64_t v, unsigned n) {
  return (v << (n & 63)) | (v >> ((64 - n) & 63));
}

/* Random123/threefry.h:190-293, Nrounds fixed at 20 (threefry.h:179).
 * Rotation schedule threefry.h:86-93, parity constant threefry.h:170-171. */
void orc_threefry2x64_20(uint64_t c0, uint64_t c1, uint64_t k0, uint64_t k1,
                         uint64_t* out0, uint64_t* out1) {
  static const unsigned rot[8] = {16, 42, 12, 31, 16, 32, 24, 21};
  uint64_t ks[3];
  ks[0] = k0;
  ks[1] = k1;
  ks[2] = UINT64_C(0x1BD11BDAA9FC1A22) ^ k0 ^ k1;

  uint64_t a = c0 + ks[0];
  uint64_t b = c1 + ks[1];
  for (unsigned r = 0; r < 20; ++r) {
    a += b;
    b = rotl64(b, rot[r & 7]);
    b ^= a;
    if ((r & 3) == 3) {
      /* key injection number s = 1..5 after rounds 4, 8, 12, 16, 20 */
      const unsigned s = (r >> 2) + 1;
      a += ks[s % 3];
      b += ks[(s + 1) % 3];
      b += s;
    }
  }
  *out0 = a;
  *out1 = b;
}

/* omp3/neutral.c:632-652 */
void orc_generate_random_numbers(uint64_t pkey, uint64_t master_key,
                                 uint64_t counter, double* rn0, double* rn1) {
  uint64_t r0, r1;
  orc_threefry2x64_20(counter, 0, pkey, master_key, &r0, &r1);
  /* 1.0 / (max_uint64 + 1.0) == 2^-64 exactly (omp3/neutral.c:647-649) */
  const double factor = 1.0 / (UINT64_C(0xFFFFFFFFFFFFFFFF) + 1.0);
  const double half_factor = 0.5 * factor;
  *rn0 = r0 * factor + half_factor;
  *rn1 = r1 * factor + half_factor;
}

/* ---- table lookup -------------------------------------------------------- */

/* omp3/neutral.c:498-517 */
double orc_microscopic_cs_for_energy(const OrcCrossSection* cs, double energy,
                                     int* cs_index) {
  const double* keys = cs->keys;
  const double* values = cs->values;

  int ind = cs->nentries / 2;
  int width = ind / 2;
  while (energy < keys[ind] || energy >= keys[ind + 1]) {
    ind += (energy < keys[ind]) ? -width : width;
    width = (width / 2 > 1) ? width / 2 : 1;
  }
  if (cs_index) {
    *cs_index = ind;
  }
  return values[ind] + ((energy - keys[ind]) / (keys[ind + 1] - keys[ind])) *
                           (values[ind + 1] - values[ind]);
}

/* ---- geometry ------------------------------------------------------------ */

/* omp3/neutral.c:423-471 */
void orc_calc_distance_to_facet(double x, double y, int pad, int x_off,
                                int y_off, double omega_x, double omega_y,
                                double speed, int particle_cellx,
                                int particle_celly, double* distance_to_facet,
                                int* x_facet, const double* edgex,
                                const double* edgey) {
  const int cellx = particle_cellx - x_off + pad;
  const int celly = particle_celly - y_off + pad;
  double u_x_inv = 1.0 / (omega_x * speed);
  double u_y_inv = 1.0 / (omega_y * speed);

  double dt_x = (omega_x >= 0.0)
                    ? ((edgex[cellx + 1]) - x) * u_x_inv
                    : ((edgex[cellx] - ORC_OPEN_BOUND_CORRECTION) - x) * u_x_inv;
  double dt_y = (omega_y >= 0.0)
                    ? ((edgey[celly + 1]) - y) * u_y_inv
                    : ((edgey[celly] - ORC_OPEN_BOUND_CORRECTION) - y) * u_y_inv;
  *x_facet = (dt_x < dt_y) ? 1 : 0;

  double mag_u0 = speed;
  if (*x_facet) {
    *distance_to_facet =
        (omega_x >= 0.0)
            ? ((edgex[cellx + 1]) - x) * mag_u0 * u_x_inv
            : ((edgex[cellx] - ORC_OPEN_BOUND_CORRECTION) - x) * mag_u0 * u_x_inv;
  } else {
    *distance_to_facet =
        (omega_y >= 0.0)
            ? ((edgey[celly + 1]) - y) * mag_u0 * u_y_inv
            : ((edgey[celly] - ORC_OPEN_BOUND_CORRECTION) - y) * mag_u0 * u_y_inv;
  }
}

/* omp3/neutral.c:474-495 */
double orc_calculate_energy_deposition(double energy, double weight,
                                       double path_length,
                                       double number_density,
                                       double microscopic_cs_absorb,
                                       double microscopic_cs_total) {
  const double average_exit_energy_absorb = 0.0;
  const double absorption_heating =
      (microscopic_cs_absorb / microscopic_cs_total) *
      average_exit_energy_absorb;
  const double average_exit_energy_scatter =
      energy * ((ORC_MASS_NO * ORC_MASS_NO + ORC_MASS_NO + 1) /
                ((ORC_MASS_NO + 1) * (ORC_MASS_NO + 1)));
  const double scattering_heating =
      (1.0 - (microscopic_cs_absorb / microscopic_cs_total)) *
      average_exit_energy_scatter;
  const double heating_response =
      (energy - scattering_heating - absorption_heating);
  return weight * path_length * (microscopic_cs_total * ORC_BARNS) *
         heating_response * number_density;
}

/* ---- injection ----------------------------------------------------------- */

/* omp3/neutral.c:560-630 */
void orc_inject_particles(int nparticles, uint64_t pid_base, int local_nx,
                          int local_ny, int pad, double local_particle_left_off,
                          double local_particle_bottom_off,
                          double local_particle_width,
                          double local_particle_height, int x_off, int y_off,
                          double dt, const double* edgex, const double* edgey,
                          double initial_energy, OrcParticles* p) {
#pragma omp parallel for
  for (int kk = 0; kk < nparticles; ++kk) {
    const uint64_t pkey = pid_base + (uint64_t)kk;
    double rn0, rn1;
    orc_generate_random_numbers(pkey, 0, 0, &rn0, &rn1);

    const double px = local_particle_left_off + rn0 * local_particle_width;
    const double py = local_particle_bottom_off + rn1 * local_particle_height;

    /* explicit search: the mesh may be non-uniform (omp3/neutral.c:588-603) */
    int cellx = 0;
    int celly = 0;
    for (int ii = 0; ii < local_nx; ++ii) {
      if (px >= edgex[ii + pad] && px < edgex[ii + pad + 1]) {
        cellx = x_off + ii;
        break;
      }
    }
    for (int ii = 0; ii < local_ny; ++ii) {
      if (py >= edgey[ii + pad] && py < edgey[ii + pad + 1]) {
        celly = y_off + ii;
        break;
      }
    }

    orc_generate_random_numbers(pkey, 0, 1, &rn0, &rn1);
    const double theta = 2.0 * M_PI * rn0;

    p->x[kk] = px;
    p->y[kk] = py;
    p->cellx[kk] = cellx;
    p->celly[kk] = celly;
    p->omega_x[kk] = cos(theta);
    p->omega_y[kk] = sin(theta);
    p->energy[kk] = initial_energy;
    p->weight[kk] = 1.0;
    p->dt_to_census[kk] = dt;
    p->mfp_to_collision[kk] = 0.0;
    p->dead[kk] = 0;
  }
}

/* ---- the history loop ---------------------------------------------------- */

/* omp3/neutral.c:408-420 */
static inline void tally_add(double* tally, int nx, int x_off, int y_off,
                             int pcellx, int pcelly,
                             double inv_ntotal_particles,
                             double energy_deposition) {
  const int cellx = pcellx - x_off;
  const int celly = pcelly - y_off;
#pragma omp atomic update
  tally[celly * nx + cellx] += energy_deposition * inv_ntotal_particles;
}

/* ---- scalar-flux tally (neutral_data.h:95) ------------------------------------
 * The reference DECLARES `double* scalar_flux_tally` and never allocates or writes
 * it in any backend, so there is no reference behaviour to restate.  This is the
 * definition the HIP path implements and is checked against here: the path-length
 * estimator of the scalar flux with the energy tally's own normalisation and flush
 * points --
 *     flux[cell] += (1/N) * sum over the track segments in the cell of weight * length
 * a segment being what collision_event / facet_event / census_event move the
 * particle by (omp3/neutral.c:227-228, 329-330, 391-392), `weight` the weight it
 * is moved with (before an absorption at the segment's end reduces it, :241), and
 * the sum flushed to the mesh where the energy deposition is (:249, :325, :397).
 * Off unless a mesh is set. */
static double* g_scalar_flux_tally = NULL;
void orc_set_scalar_flux_tally(double* tally) { g_scalar_flux_tally = tally; }

/* census events of the most recent orc_solve_transport_2d call (bookkeeping for
 * the particle-steps metric; the reference does not count them) */
static uint64_t g_last_census = 0;
uint64_t orc_last_census(void) { return g_last_census; }

/* omp3/neutral.c:19-206 */
uint64_t orc_solve_transport_2d(int nx, int ny, int global_nx, int global_ny,
                                uint64_t master_key, int pad, int x_off,
                                int y_off, double dt, int ntotal_particles,
                                int nparticles_to_process, uint64_t pid_base,
                                OrcParticles* p, const double* density,
                                const double* edgex, const double* edgey,
                                const OrcCrossSection* cs_scatter,
                                const OrcCrossSection* cs_absorb,
                                double* energy_deposition_tally,
                                uint64_t* facets, uint64_t* collisions) {
  (void)ny;
  if (!nparticles_to_process) {
    /* omp3/neutral.c:30-33 */
    printf("Out of particles\n");
    return 0;
  }

  uint64_t nfacets = 0;
  uint64_t ncollisions = 0;
  uint64_t nprocessed = 0;
  uint64_t ncensus = 0;

  /* omp3/neutral.c:64-78 is a hand-written static block partition; schedule
   * (static) over the same index range assigns the same contiguous blocks. */
#pragma omp parallel for schedule(static) \
    reduction(+ : nfacets, ncollisions, nprocessed, ncensus)
  for (int pid = 0; pid < nparticles_to_process; ++pid) {
    if (p->dead[pid]) {
      continue; /* omp3/neutral.c:91-93 */
    }
    nprocessed++;

    const uint64_t pkey = pid_base + (uint64_t)pid; /* omp3/neutral.c:89 */

    double px = p->x[pid];
    double py = p->y[pid];
    double omega_x = p->omega_x[pid];
    double omega_y = p->omega_y[pid];
    double energy = p->energy[pid];
    double weight = p->weight[pid];
    double dt_to_census = p->dt_to_census[pid];
    double mfp_to_collision = p->mfp_to_collision[pid];
    int pcellx = p->cellx[pid];
    int pcelly = p->celly[pid];
    int dead = 0;

    int x_facet = 0;
    int scatter_cs_index = -1;
    int absorb_cs_index = -1;
    double cell_mfp = 0.0;

    /* omp3/neutral.c:103-105 */
    int cellx = pcellx - x_off + pad;
    int celly = pcelly - y_off + pad;
    double local_density = density[celly * (nx + 2 * pad) + cellx];

    /* omp3/neutral.c:108-117 */
    double microscopic_cs_scatter =
        orc_microscopic_cs_for_energy(cs_scatter, energy, &scatter_cs_index);
    double microscopic_cs_absorb =
        orc_microscopic_cs_for_energy(cs_absorb, energy, &absorb_cs_index);
    double number_density = (local_density * ORC_AVOGADROS / ORC_MOLAR_MASS);
    double macroscopic_cs_scatter =
        number_density * microscopic_cs_scatter * ORC_BARNS;
    double macroscopic_cs_absorb =
        number_density * microscopic_cs_absorb * ORC_BARNS;
    double speed = sqrt((2.0 * energy * ORC_eV_TO_J) / ORC_PARTICLE_MASS);
    double energy_deposition = 0.0;
    double track_length = 0.0; /* weight * path length not yet tallied (scalar flux) */
    double* const flux_tally = g_scalar_flux_tally;

    const double inv_ntotal_particles = 1.0 / (double)ntotal_particles;

    uint64_t counter = 0;
    double rn0, rn1;

    /* omp3/neutral.c:127-131, initial == 1 always (omp3/neutral.c:35-36) */
    dt_to_census = dt;
    orc_generate_random_numbers(pkey, master_key, counter++, &rn0, &rn1);
    mfp_to_collision = -log(rn0) / macroscopic_cs_scatter;

    /* omp3/neutral.c:134-197 */
    while (dt_to_census > 0.0) {
      cell_mfp = 1.0 / (macroscopic_cs_scatter + macroscopic_cs_absorb);

      double distance_to_facet = 0.0;
      orc_calc_distance_to_facet(px, py, pad, x_off, y_off, omega_x, omega_y,
                                 speed, pcellx, pcelly, &distance_to_facet,
                                 &x_facet, edgex, edgey);

      const double distance_to_collision = mfp_to_collision * cell_mfp;
      const double distance_to_census = speed * dt_to_census;

      if (distance_to_collision < distance_to_facet &&
          distance_to_collision < distance_to_census) {
        /* ---- collision_event, omp3/neutral.c:209-300 ---- */
        ncollisions++;

        energy_deposition += orc_calculate_energy_deposition(
            energy, weight, distance_to_collision, number_density,
            microscopic_cs_absorb,
            microscopic_cs_scatter + microscopic_cs_absorb);
        track_length += weight * distance_to_collision;

        px += distance_to_collision * omega_x;
        py += distance_to_collision * omega_y;

        const double p_absorb =
            macroscopic_cs_absorb /
            (macroscopic_cs_scatter + macroscopic_cs_absorb);

        double rc0, rc1;
        orc_generate_random_numbers(pkey, master_key, counter++, &rc0, &rc1);

        if (rc0 < p_absorb) {
          /* absorption, omp3/neutral.c:237-252 */
          weight *= (1.0 - p_absorb);
          if (energy < ORC_MIN_ENERGY_OF_INTEREST) {
            dead = 1;
            tally_add(energy_deposition_tally, nx, x_off, y_off, pcellx,
                      pcelly, inv_ntotal_particles, energy_deposition);
            energy_deposition = 0.0;
            if (flux_tally) {
              tally_add(flux_tally, nx, x_off, y_off, pcellx, pcelly,
                        inv_ntotal_particles, track_length);
              track_length = 0.0;
            }
            break; /* PARTICLE_DEAD, omp3/neutral.c:165-167 */
          }
        } else {
          /* elastic scatter, omp3/neutral.c:253-282 */
          const double mu_cm = 1.0 - 2.0 * rc1;
          const double e_new =
              energy *
              (ORC_MASS_NO * ORC_MASS_NO + 2.0 * ORC_MASS_NO * mu_cm + 1.0) /
              ((ORC_MASS_NO + 1.0) * (ORC_MASS_NO + 1.0));
          double cos_theta =
              0.5 * ((ORC_MASS_NO + 1.0) * sqrt(e_new / energy) -
                     (ORC_MASS_NO - 1.0) * sqrt(energy / e_new));
          const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
          const double omega_x_new = (omega_x * cos_theta - omega_y * sin_theta);
          const double omega_y_new = (omega_x * sin_theta + omega_y * cos_theta);
          omega_x = omega_x_new;
          omega_y = omega_y_new;
          energy = e_new;
        }

        /* omp3/neutral.c:284-297 */
        microscopic_cs_scatter =
            orc_microscopic_cs_for_energy(cs_scatter, energy, &scatter_cs_index);
        microscopic_cs_absorb =
            orc_microscopic_cs_for_energy(cs_absorb, energy, &absorb_cs_index);
        number_density = (local_density * ORC_AVOGADROS / ORC_MOLAR_MASS);
        macroscopic_cs_scatter =
            number_density * microscopic_cs_scatter * ORC_BARNS;
        macroscopic_cs_absorb =
            number_density * microscopic_cs_absorb * ORC_BARNS;

        orc_generate_random_numbers(pkey, master_key, counter++, &rn0, &rn1);
        mfp_to_collision = -log(rn0) / macroscopic_cs_scatter;
        dt_to_census -= distance_to_collision / speed;
        speed = sqrt((2.0 * energy * ORC_eV_TO_J) / ORC_PARTICLE_MASS);
      } else if (distance_to_facet < distance_to_census) {
        /* ---- facet_event, omp3/neutral.c:303-380 ---- */
        nfacets++;

        mfp_to_collision -= (distance_to_facet / cell_mfp);
        dt_to_census -= (distance_to_facet / speed);

        energy_deposition += orc_calculate_energy_deposition(
            energy, weight, distance_to_facet, number_density,
            microscopic_cs_absorb,
            microscopic_cs_scatter + microscopic_cs_absorb);

        track_length += weight * distance_to_facet;

        tally_add(energy_deposition_tally, nx, x_off, y_off, pcellx, pcelly,
                  inv_ntotal_particles, energy_deposition);
        energy_deposition = 0.0;
        if (flux_tally) {
          tally_add(flux_tally, nx, x_off, y_off, pcellx, pcelly,
                    inv_ntotal_particles, track_length);
          track_length = 0.0;
        }

        px += distance_to_facet * omega_x;
        py += distance_to_facet * omega_y;

        if (x_facet) {
          if (omega_x > 0.0) {
            if (pcellx >= (global_nx - 1)) {
              omega_x = -(omega_x);
            } else {
              pcellx++;
            }
          } else if (omega_x < 0.0) {
            if (pcellx <= 0) {
              omega_x = -(omega_x);
            } else {
              pcellx--;
            }
          }
        } else {
          if (omega_y > 0.0) {
            if (pcelly >= (global_ny - 1)) {
              omega_y = -(omega_y);
            } else {
              pcelly++;
            }
          } else if (omega_y < 0.0) {
            if (pcelly <= 0) {
              omega_y = -(omega_y);
            } else {
              pcelly--;
            }
          }
        }

        /* omp3/neutral.c:371-377 */
        cellx = pcellx - x_off;
        celly = pcelly - y_off;
        local_density = density[celly * nx + cellx];
        number_density = (local_density * ORC_AVOGADROS / ORC_MOLAR_MASS);
        macroscopic_cs_scatter =
            number_density * microscopic_cs_scatter * ORC_BARNS;
        macroscopic_cs_absorb =
            number_density * microscopic_cs_absorb * ORC_BARNS;
      } else {
        /* ---- census_event, omp3/neutral.c:383-405 ---- */
        px += distance_to_census * omega_x;
        py += distance_to_census * omega_y;
        mfp_to_collision -= (distance_to_census / cell_mfp);
        energy_deposition += orc_calculate_energy_deposition(
            energy, weight, distance_to_census, number_density,
            microscopic_cs_absorb,
            microscopic_cs_scatter + microscopic_cs_absorb);
        track_length += weight * distance_to_census;
        tally_add(energy_deposition_tally, nx, x_off, y_off, pcellx, pcelly,
                  inv_ntotal_particles, energy_deposition);
        if (flux_tally) {
          tally_add(flux_tally, nx, x_off, y_off, pcellx, pcelly,
                    inv_ntotal_particles, track_length);
        }
        dt_to_census = 0.0;
        ncensus++;
        break;
      }
    }

    p->x[pid] = px;
    p->y[pid] = py;
    p->omega_x[pid] = omega_x;
    p->omega_y[pid] = omega_y;
    p->energy[pid] = energy;
    p->weight[pid] = weight;
    p->dt_to_census[pid] = dt_to_census;
    p->mfp_to_collision[pid] = mfp_to_collision;
    p->cellx[pid] = pcellx;
    p->celly[pid] = pcelly;
    p->dead[pid] = dead;
  }

  /* omp3/neutral.c:202-205 */
  *facets += nfacets;
  *collisions += ncollisions;
  g_last_census = ncensus;
  return nprocessed;
}

/* omp3/neutral.c:524-527 */
double orc_sum_tally(int nx, int ny, const double* energy_deposition_tally) {
  double local_energy_tally = 0.0;
  for (int ii = 0; ii < nx * ny; ++ii) {
    local_energy_tally += energy_deposition_tally[ii];
  }
  return local_energy_tally;
}

int orc_num_threads(void) {
  int n = 1;
#pragma omp parallel
  {
#pragma omp master
    n = omp_get_num_threads();
  }
  return n;
}

void orc_set_num_threads(int n) { omp_set_num_threads(n); }
