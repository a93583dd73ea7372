/* Rank layer of the host library, exercised by N real processes on the CPU
 * (tests/test_comms_ranks.py starts them with RANK / WORLD_SIZE / MASTER_PORT):
 * rendezvous, the reference-facing hooks (initialise_mpi, barrier, reduce_all_*),
 * the byte broadcast that carries the RCCL id, the array all-reduce that carries
 * the tally when RCCL is not used, and the particle shards.
 * Prints one line "rank R of N ok" and exits 0, or says what went wrong. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "comms.h"
#include "mesh.h"
#include "shared.h"

#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      fprintf(stderr, "rank %d: check failed: %s (line %d)\n", comms_rank(), #cond, __LINE__); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

int main(int argc, char** argv) {
  int rank = -1, nranks = -1;
  initialise_mpi(argc, argv, &rank, &nranks);
  initialise_devices(rank); /* host flavour: nothing to bind */
  CHECK(rank == comms_rank() && nranks == comms_nranks());
  CHECK(rank >= 0 && rank < nranks);
  barrier();

  /* scalar reductions of main.c / omp3 validate */
  const double n = (double)nranks;
  CHECK(reduce_all_sum((double)(rank + 1)) == n * (n + 1.0) / 2.0);
  CHECK(reduce_all_min((double)(rank + 1)) == 1.0);
  CHECK(reduce_all_max((double)(rank + 1)) == n);

  /* broadcast of opaque bytes from rank 0 (the RCCL unique id travels this way) */
  unsigned char id[128];
  for (int i = 0; i < 128; ++i) id[i] = (unsigned char)(rank == 0 ? (i * 7 + 3) : 0xEE);
  comms_bcast_bytes(id, sizeof(id));
  for (int i = 0; i < 128; ++i) CHECK(id[i] == (unsigned char)(i * 7 + 3));

  /* tally-sized array all-reduce: same bits on every rank, equal to the sum in rank order */
  const size_t cells = 400 * 400;
  double* t = (double*)malloc(sizeof(double) * cells);
  double* want = (double*)malloc(sizeof(double) * cells);
  for (size_t i = 0; i < cells; ++i) {
    want[i] = 0.0;
    for (int r = 0; r < nranks; ++r) want[i] += 1.0e-7 * (double)(i % 977) / (double)(r + 3);
    t[i] = 1.0e-7 * (double)(i % 977) / (double)(rank + 3);
  }
  comms_allreduce_f64(t, cells, COMMS_SUM);
  CHECK(memcmp(t, want, sizeof(double) * cells) == 0);
  uint64_t counts[4] = {(uint64_t)rank, 1u, 1ull << 40, 7u};
  comms_allreduce_u64(counts, 4, COMMS_SUM);
  CHECK(counts[0] == (uint64_t)(nranks * (nranks - 1) / 2) && counts[1] == (uint64_t)nranks);
  CHECK(counts[2] == ((uint64_t)nranks << 40));

  /* personalised exchange (the particle exchange of the decomposed-mesh mode): rank s
   * hands (s + 2 d + 1) records of 80 bytes to rank d, each stamped with (s, d, k) */
  {
    uint64_t* m = (uint64_t*)calloc((size_t)nranks * nranks, sizeof(uint64_t));
    size_t out = 0, in = 0;
    for (int d = 0; d < nranks; ++d) {
      m[(size_t)rank * nranks + d] = (uint64_t)(rank + 2 * d + 1) * 80u;
      out += (size_t)m[(size_t)rank * nranks + d];
    }
    comms_allreduce_u64(m, (size_t)nranks * nranks, COMMS_SUM);
    for (int s2 = 0; s2 < nranks; ++s2) in += (size_t)m[(size_t)s2 * nranks + rank];
    int* sb = (int*)calloc(out / 4 + 1, 4);
    int* rb = (int*)calloc(in / 4 + 1, 4);
    size_t at = 0;
    for (int d = 0; d < nranks; ++d) {
      for (int k = 0; k < rank + 2 * d + 1; ++k, at += 20) {
        sb[at] = rank; sb[at + 1] = d; sb[at + 2] = k;
      }
    }
    comms_alltoallv(sb, rb, m);
    at = 0;
    for (int s2 = 0; s2 < nranks; ++s2) {
      for (int k = 0; k < s2 + 2 * rank + 1; ++k, at += 20) {
        CHECK(rb[at] == s2 && rb[at + 1] == rank && rb[at + 2] == k);
      }
    }
    CHECK(at * 4 == in);
    free(m); free(sb); free(rb);
  }

  /* shards: contiguous, disjoint, covering, the first total % nranks one longer */
  const long long total = 100000007LL;
  long long next = 0;
  for (int r = 0; r < nranks; ++r) {
    long long first, count;
    comms_shard_range(total, r, nranks, &first, &count);
    CHECK(first == next);
    CHECK(count == total / nranks + (r < total % nranks ? 1 : 0));
    next = first + count;
  }
  CHECK(next == total);

  barrier();
  printf("rank %d of %d ok\n", rank, nranks);
  finalise_comms();
  free(t);
  free(want);
  return 0;
}
