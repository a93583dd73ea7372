/*
 * neutral_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C + OpenMP) of the over-particle history loop of
 * UoB-HPC/neutral's `omp3` kernel set.  It exists only as the parity checker
 * for the HIP path and as the `cpu_baseline` leg of bench.py.  Nothing under
 * neutral_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line it restates (paths relative
 * to the reference tree).  Data layout is the reference's `-DSoA` Particle
 * (neutral_data.h:45-61) so that oracle and HIP states compare array by array;
 * the arithmetic and its order follow omp3/neutral.c, which is AoS -- layout
 * does not enter any result.
 *
 * Pinning (see DESIGN.md "Oracle"): Threefry2x64-20 is checked bit-for-bit
 * against the reference's own Random123/threefry.h compiled in place
 * (oracle/_ref); the full loop is checked against the three known answers of
 * problems/neutral.tests and against the omp3 event counts / tallies recorded
 * in BASELINE.md section 2.  omp3/neutral.c itself is unbuildable here (it
 * needs the absent parent `arch` project), so no compiled omp3 is used.
 */
#ifndef NEUTRAL_ORACLE_H
#define NEUTRAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* neutral_data.h:17-24 */
#define ORC_eV_TO_J 1.60217646e-19
#define ORC_AVOGADROS 6.02214085774e23
#define ORC_BARNS 1.0e-28
#define ORC_PARTICLE_MASS 1.674927471213e-27
#define ORC_MASS_NO 1.0e2
#define ORC_MOLAR_MASS 1.0e-2
#define ORC_MIN_ENERGY_OF_INTEREST 1.0e0
#define ORC_OPEN_BOUND_CORRECTION 1.0e-13

/* neutral_data.h:38-43 */
typedef struct {
  double* keys;
  double* values;
  int nentries;
} OrcCrossSection;

/* neutral_data.h:48-61 (SoA flavour) */
typedef struct {
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
} OrcParticles;

/* Random123/threefry.h:190-293 with Nrounds = 20 (threefry.h:179) */
void orc_threefry2x64_20(uint64_t c0, uint64_t c1, uint64_t k0, uint64_t k1,
                         uint64_t* out0, uint64_t* out1);

/* omp3/neutral.c:632-652 */
void orc_generate_random_numbers(uint64_t pkey, uint64_t master_key,
                                 uint64_t counter, double* rn0, double* rn1);

/* omp3/neutral.c:498-517; *cs_index receives the bracket index found */
double orc_microscopic_cs_for_energy(const OrcCrossSection* cs, double energy,
                                     int* cs_index);

/* omp3/neutral.c:423-471 */
void orc_calc_distance_to_facet(double x, double y, int pad, int x_off,
                                int y_off, double omega_x, double omega_y,
                                double speed, int particle_cellx,
                                int particle_celly, double* distance_to_facet,
                                int* x_facet, const double* edgex,
                                const double* edgey);

/* omp3/neutral.c:474-495 */
double orc_calculate_energy_deposition(double energy, double weight,
                                       double path_length,
                                       double number_density,
                                       double microscopic_cs_absorb,
                                       double microscopic_cs_total);

/* omp3/neutral.c:560-630.  Fills particles [0, nparticles) whose RNG key is
 * pid_base + index (pid_base = 0 reproduces the reference; a non-zero base is
 * the particle-shard extension of SURVEY section 8(e)).  Arrays are caller
 * allocated. */
void orc_inject_particles(int nparticles, uint64_t pid_base, int local_nx,
                          int local_ny, int pad, double local_particle_left_off,
                          double local_particle_bottom_off,
                          double local_particle_width,
                          double local_particle_height, int x_off, int y_off,
                          double dt, const double* edgex, const double* edgey,
                          double initial_energy, OrcParticles* p);

/* omp3/neutral.c:19-206 (solve_transport_2d -> handle_particles, initial=1).
 * Returns the number of particles processed (the "Particles  N" line);
 * *facets and *collisions are incremented as at omp3/neutral.c:202-203. */
uint64_t orc_solve_transport_2d(int nx, int ny, int global_nx, int global_ny,
                                uint64_t master_key, int pad, int x_off,
                                int y_off, double dt, int ntotal_particles,
                                int nparticles, uint64_t pid_base,
                                OrcParticles* p, const double* density,
                                const double* edgex, const double* edgey,
                                const OrcCrossSection* cs_scatter,
                                const OrcCrossSection* cs_absorb,
                                double* energy_deposition_tally,
                                uint64_t* facets, uint64_t* collisions);

/* Scalar-flux tally (declared, never written, in the reference: neutral_data.h:95):
 * when set (nx*ny doubles), orc_solve_transport_2d also accumulates the path-length
 * estimator sum(weight * segment length) / N per cell, flushed where the energy
 * deposition is.  NULL (default) turns it off.  See neutral_oracle.c. */
void orc_set_scalar_flux_tally(double* tally);

/* histories of the most recent orc_solve_transport_2d call that ended in a
 * census event (not counted by the reference; needed for particle-steps) */
uint64_t orc_last_census(void);

/* omp3/neutral.c:524-527: serial sum of the tally mesh */
double orc_sum_tally(int nx, int ny, const double* energy_deposition_tally);

/* number of OpenMP threads the loop will use */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
