/*
 * shared.h -- host layer of neutral_amd: the minimal `arch`-compatible surface
 * the reference driver expects one directory above itself.
 *
 * UoB-HPC/neutral is written to live inside the parent project UoB-HPC/arch
 * (reference README.md:10-17) and includes ../shared.h, ../mesh.h, ../comms.h,
 * ../params.h, ../profiler.h, ../shared_data.h (main.c:1-5, neutral_data.c:2-4,
 * omp3/neutral.c:2-5).  That project is not part of the reference tree, so this
 * directory supplies, from scratch, exactly the symbols those translation
 * units use (list: SURVEY.md section 8(b)).  Semantics that the reference tree
 * does not pin are documented where they are implemented (host.c).
 */
#ifndef NEUTRAL_AMD_HOST_SHARED_H
#define NEUTRAL_AMD_HOST_SHARED_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "profiler.h" /* omp3-style kernel sets use START_PROFILING via shared.h */

#ifdef __cplusplus
extern "C" {
#endif

#define MASTER 0
#define NNEIGHBOURS 6
#define EDGE (-1)
#define GB (1024.0 * 1024.0 * 1024.0)
#define MAX_KEYS 40
#define MAX_STR_LEN 1024

/* handle_boundary_2d selectors (main.c:70-71) */
enum { NO_INVERT, INVERT_X, INVERT_Y };
enum { NO_PACK, PACK };
/* copy_buffer direction (neutral_data.c:59-62): RECV = device -> host */
enum { SEND, RECV };

#ifndef min
#define min(a, b) (((a) < (b)) ? (a) : (b))
#endif
#ifndef max
#define max(a, b) (((a) > (b)) ? (a) : (b))
#endif

/* fatal error: print and exit, as every TERMINATE site in the reference
 * expects (main.c:22, neutral_data.c:35,126, omp3/neutral.c:572) */
#define TERMINATE(...)                                              \
  do {                                                              \
    fprintf(stderr, __VA_ARGS__);                                   \
    fprintf(stderr, "%s:%d\n", __FILE__, __LINE__);                 \
    exit(EXIT_FAILURE);                                             \
  } while (0)

/* relative tolerance check used by validate (omp3/neutral.c:549) */
int within_tolerance(const double expected, const double result,
                     const double tolerance);

void initialise_devices(int rank);

/* ---- allocation hooks ----------------------------------------------------
 * Which memory space these return is decided by the object linked in
 * (neutral_data.c:97-105,168-169): alloc_host.c -> malloc'd host memory (CPU
 * kernel sets), libneutral_hip.so -> HBM (hipMalloc).  All return the number
 * of bytes allocated and zero-fill the buffer. */
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
/* copies len doubles between *src and *dst; RECV: device->host, SEND:
 * host->device */
void copy_buffer(const size_t len, double** src, double** dst, int send);
void copy_int_buffer(const size_t len, int** src, int** dst, int send);
/* uploads *src (host, then freed) into a new device buffer stored in *dst */
void move_host_buffer_to_device(const size_t len, double** src, double** dst);

#ifdef __cplusplus
}
#endif
#endif
