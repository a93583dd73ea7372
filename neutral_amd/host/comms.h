/*
 * comms.h -- rank-level communication surface (main.c:62,64,70-71,75,112;
 * omp3/neutral.c:530).  The reference ships single-rank (main.c:42-43, MPI blocks
 * are `#if 0`: neutral_data.h:10-14) but leaves rank and rank count to
 * initialise_mpi (main.c:62): this layer runs one process per GPU on one node,
 * particles sharded over the ranks, the mesh replicated (comms_ranks.c).
 */
#ifndef NEUTRAL_AMD_HOST_COMMS_H
#define NEUTRAL_AMD_HOST_COMMS_H

#include "mesh.h"
#include "shared.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rank and rank count from the launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK,
 * MASTER_ADDR, MASTER_PORT); one rank when they are absent */
void initialise_mpi(int argc, char** argv, int* rank, int* nranks);

/* the rank layer itself (comms_ranks.c) */
enum { COMMS_SUM, COMMS_MIN, COMMS_MAX };
void comms_start_from_env(void); /* idempotent; rank 0 listens, the others connect */
void comms_stop(void);
int comms_rank(void);
int comms_nranks(void);
int comms_local_rank(void);
/* contiguous particle-id range of a rank (the OpenMP static split of
 * omp3/neutral.c:64-74 over ranks) */
void comms_shard_range(long long total, int rank, int nranks, long long* first,
                       long long* count);
void comms_bcast_bytes(void* buf, size_t n); /* from rank 0 */
/* element-wise over the ranks, result everywhere, same bits everywhere (host arrays) */
void comms_allreduce_f64(double* buf, size_t n, int op);
void comms_allreduce_u64(uint64_t* buf, size_t n, int op);
/* personalised exchange of bytes: rank s hands matrix[s * nranks + d] bytes to rank d
 * (send buffer ordered by d, receive buffer ordered by s); the matrix is known on
 * every rank */
void comms_alltoallv(const void* sendbuf, void* recvbuf, const uint64_t* matrix);
void comms_barrier(void);
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
