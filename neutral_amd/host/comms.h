/*
 * comms.h -- rank-level communication surface (main.c:62,64,70-71,75,112;
 * omp3/neutral.c:530).  The reference runs single-rank (main.c:42-43, MPI
 * blocks are `#if 0`: neutral_data.h:10-14), so these are the one-rank forms.
 */
#ifndef NEUTRAL_AMD_HOST_COMMS_H
#define NEUTRAL_AMD_HOST_COMMS_H

#include "mesh.h"
#include "shared.h"

#ifdef __cplusplus
extern "C" {
#endif

void initialise_mpi(int argc, char** argv, int* rank, int* nranks);
void initialise_comms(Mesh* mesh);
void finalise_comms(void);
void barrier(void);
double reduce_all_sum(double local_val);
double reduce_all_min(double local_val);
double reduce_all_max(double local_val);
void handle_boundary_2d(const int nx, const int ny, Mesh* mesh, double* arr,
                        const int invert, const int pack);
/* VisIt output is out of scope: prints a notice */
void write_all_ranks_to_visit(const int global_nx, const int global_ny,
                              const int local_nx, const int local_ny,
                              const int pad, const int x_off, const int y_off,
                              const int rank, const int nranks,
                              int* neighbours, double* local_arr,
                              const char* name, const int tt,
                              const double elapsed_sim_time);

#ifdef __cplusplus
}
#endif
#endif
