/*
 * neutral_abi_exchange.hip -- what leaves the device at the end of a batch of launches: the
 * step's results published into pinned host memory, the tally exchange between the ranks of a
 * sharded run (one all-reduce per step, the event counters riding on a second one), and the
 * particle exchange between the blocks of a decomposed mesh.  Host code and three small kernels.
 */
#include "neutral_abi_state.h"

namespace neutral_abi {

/* the step's results, into the pinned block the host reads after its wait (one workgroup) */
__global__ void publish_results_kernel(const neutral::StepCounters* counters,
                                       const unsigned long long* check, const unsigned* ctrl,
                                       const unsigned long long* words, StepResults* out) {
  const unsigned t = threadIdx.x;
  const unsigned* c32 = (const unsigned*)counters;
  unsigned* o32 = (unsigned*)out->counters;
  for (unsigned i = t; i < 2 * sizeof(neutral::StepCounters) / 4; i += blockDim.x) {
    o32[i] = c32[i];
  }
  if (t < 8) out->check[t] = check[t];
  if (t < 16) out->ctrl[t] = ctrl ? ctrl[t] : 0u;
  if (t < (unsigned)kStepWords) out->words[t] = words ? words[t] : 0ull;
}

__global__ void pack_step_words_kernel(const neutral::StepCounters* c, const unsigned long long* check,
                                       const unsigned* ctrl, unsigned long long* w) {
  if (threadIdx.x != 0) {
    return;
  }
  for (int k = 0; k < 2; ++k) {
    w[kWordCounters + 4 * k + 0] = c[k].nprocessed;
    w[kWordCounters + 4 * k + 1] = c[k].nfacets;
    w[kWordCounters + 4 * k + 2] = c[k].ncollisions;
    w[kWordCounters + 4 * k + 3] = c[k].ncensus;
  }
  w[kWordRequeued] = c[0].nrequeued + c[1].nrequeued;
  w[kWordCollidePasses] = c[0].ncollide_passes + c[1].ncollide_passes;
  w[kWordTurnedDown] = check[0] ? 1ull : 0ull;
  w[kWordMigrants] = ctrl ? ctrl[4] : 0u;
  w[kWordQueued] = ctrl ? ctrl[2] : 0u;
  w[kWordAborted] = (unsigned long long)c[0].aborted + c[1].aborted;
  w[kWordRanks] = 1ull;
  w[kWordSteals] = c[0].nsteals + c[1].nsteals;
  w[kWordStealsRefused] = c[0].steal_refused + c[1].steal_refused;
  w[kWordWeightedWaves] = c[0].nweighted + c[1].nweighted;
  w[kStepWords - 2] = 0ull;
  w[kStepWords - 1] = 0ull;
}

__global__ void add_step_tally_kernel(double* __restrict__ tally, const double* __restrict__ step,
                                      size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    tally[i] += step[i];
  }
}

/* End of a timestep with several ranks: ONE all-reduce of the step's tally
 * contributions (sum, f64, nx*ny) on the kernels' stream; the sum joins the caller's
 * mesh, which then holds the same global tally on every rank.  The step buffer is
 * cleared again, so a step that needs more stream passes than were enqueued simply
 * exchanges what those add. */
void exchange_step(const neutral::SolveArgs& a, double* tally, bool tiled) {
  /* on a stream of its own, after the step's kernels (g.ev_stop) and BESIDE the write-back
   * of the records that the caller enqueues next on its own stream; finish_exchange() joins */
  hipStream_t xs = g.comm_stream;
  HIP_CHECK(hipStreamWaitEvent(xs, g.ev_stop, 0));
  HIP_CHECK(hipEventRecord(g.ev_exchange_begins, xs));
  const size_t ncells = (size_t)a.nx * (size_t)a.ny;
  hipLaunchKernelGGL(pack_step_words_kernel, dim3(1), dim3(64), 0, xs, g.d_counters, g.d_check,
                     tiled ? (const unsigned*)g.tiled.ctrl : (const unsigned*)nullptr, g.d_words);
  HIP_CHECK(hipGetLastError());
  neutral::comm_allreduce_sum(g.d_words, (size_t)kStepWords, false, xs);
  neutral::comm_allreduce_sum(a.tally, ncells, true, xs);
  hipLaunchKernelGGL(add_step_tally_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0,
                     xs, tally, (const double*)a.tally, ncells);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemsetAsync(a.tally, 0, sizeof(double) * ncells, xs));
  if (g.flux_tally) { /* the scalar-flux mesh travels the same way */
    neutral::comm_allreduce_sum(a.flux_tally, ncells, true, xs);
    hipLaunchKernelGGL(add_step_tally_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0,
                       xs, g.flux_tally, (const double*)a.flux_tally, ncells);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemsetAsync(a.flux_tally, 0, sizeof(double) * ncells, xs));
  }
  HIP_CHECK(hipEventRecord(g.ev_exchanged, xs));
}

/* the caller's stream goes on only when the exchange is done (the step buffers are reused) */
void finish_exchange() { HIP_CHECK(hipStreamWaitEvent(g.stream, g.ev_exchanged, 0)); }

/* enqueues the publication of the batch's results; fetch_results() after the wait */
void publish_results(bool tiled, bool with_words) {
  hipLaunchKernelGGL(publish_results_kernel, dim3(1), dim3(64), 0, g.stream, g.d_counters, g.d_check,
                     tiled ? (const unsigned*)g.tiled.ctrl : (const unsigned*)nullptr,
                     with_words ? (const unsigned long long*)g.d_words
                                : (const unsigned long long*)nullptr,
                     g.d_results);
  HIP_CHECK(hipGetLastError());
}

void fetch_results(neutral::StepCounters* hc, unsigned long long* check, unsigned* ctrl,
                   unsigned long long* words) {
  const StepResults& r = *g.h_results;
  memcpy(hc, r.counters, sizeof(r.counters));
  if (check) memcpy(check, r.check, sizeof(r.check));
  if (ctrl) memcpy(ctrl, r.ctrl, sizeof(r.ctrl));
  if (words) memcpy(words, r.words, sizeof(r.words));
}

/* Decomposed mesh, one round: this rank's emigrants (records of t.rec_out marked
 * kRecEmigrate) go to the ranks that own the cells they crossed into; what arrives
 * is appended behind the a.nparticles records already here, as migrants.  Returns the
 * number of arrivals.  Collective over the ranks. */
int exchange_particles(const neutral::SolveArgs& a, neutral::TiledArgs& t) {
  const int n = neutral::comm_nranks();
  const int me = neutral::comm_rank();
  if (n > 64) {
    fprintf(stderr, "libneutral_hip: the decomposed-mesh exchange handles up to 64 ranks.\n");
    exit(EXIT_FAILURE);
  }
  unsigned* d_counts = g.d_exchange;
  unsigned* d_offsets = g.d_exchange + 64;
  unsigned* d_cursor = g.d_exchange + 128;
  HIP_CHECK(hipMemsetAsync(g.d_exchange, 0, sizeof(unsigned) * 192, g.stream));
  HIP_CHECK(neutral::launch_emigrant_count(t, a.nparticles, g.domain, d_counts, g.stream));
  unsigned counts[64];
  HIP_CHECK(hipMemcpyAsync(counts, d_counts, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToHost,
                           g.stream));
  wait_for_stream();
  uint64_t matrix[64 * 64];
  memset(matrix, 0, sizeof(uint64_t) * (size_t)n * n);
  unsigned offsets[64];
  size_t out = 0;
  for (int d = 0; d < n; ++d) {
    offsets[d] = (unsigned)out;
    out += counts[d];
    matrix[(size_t)me * n + d] = (uint64_t)counts[d] * sizeof(neutral::ParticleRec);
  }
  if (counts[me] != 0) {
    fprintf(stderr, "libneutral_hip: rank %d: %u emigrants are bound for their own rank (the "
                    "decomposition given to neutral_hip_set_decomposition does not match the "
                    "mesh blocks passed to solve_transport_2d).\n", me, counts[me]);
    exit(EXIT_FAILURE);
  }
  comms_allreduce_u64(matrix, (size_t)n * n, COMMS_SUM);
  g.host_collectives++;
  size_t in = 0;
  for (int s2 = 0; s2 < n; ++s2) {
    in += (size_t)(matrix[(size_t)s2 * n + me] / sizeof(neutral::ParticleRec));
  }
  if (out > g.send_capacity) {
    if (g.d_send) HIP_CHECK(hipFree(g.d_send));
    g.send_capacity = out + out / 2 + 1024;
    HIP_CHECK(hipMalloc((void**)&g.d_send, sizeof(neutral::ParticleRec) * g.send_capacity));
  }
  if (in > g.recv_capacity) {
    if (g.d_recv) HIP_CHECK(hipFree(g.d_recv));
    g.recv_capacity = in + in / 2 + 1024;
    HIP_CHECK(hipMalloc((void**)&g.d_recv, sizeof(neutral::ParticleRec) * g.recv_capacity));
  }
  if ((size_t)g.tiled_particles > g.free_slots_capacity) {
    if (g.d_free_slots) HIP_CHECK(hipFree(g.d_free_slots));
    g.free_slots_capacity = (size_t)g.tiled_particles;
    HIP_CHECK(hipMalloc((void**)&g.d_free_slots, sizeof(unsigned) * g.free_slots_capacity));
  }
  /* the free list's length lives on the device next to the other exchange words */
  unsigned* d_nfree = g.d_exchange + 193;
  const unsigned nfree_now = (unsigned)g.free_count;
  HIP_CHECK(hipMemcpyAsync(d_nfree, &nfree_now, sizeof(unsigned), hipMemcpyHostToDevice, g.stream));
  HIP_CHECK(hipMemcpyAsync(d_offsets, offsets, sizeof(unsigned) * (size_t)n, hipMemcpyHostToDevice,
                           g.stream));
  HIP_CHECK(neutral::launch_emigrant_pack(t, a.nparticles, g.domain, d_offsets, d_cursor, g.d_send,
                                          g.d_free_slots, d_nfree, g.stream));
  g.free_count += (int)out;
  g.exchange_rounds++;
  g.emigrants += (unsigned long long)out;
  neutral::comm_exchange_bytes(g.d_send, g.d_recv, matrix, g.stream);
  g.host_syncs++;
  /* arrivals take the slots emigrants left first, then slots behind the records */
  const int reuse = ((int)in < g.free_count) ? (int)in : g.free_count;
  const int grow = (int)in - reuse;
  if ((size_t)a.nparticles + (size_t)grow > (size_t)g.tiled_particles) {
    fprintf(stderr, "libneutral_hip: rank %d: %zu particles arrive but the store is full (%d "
                    "slots).\n", me, in, g.tiled_particles);
    exit(EXIT_FAILURE);
  }
  HIP_CHECK(neutral::launch_immigrant_append(t, g.d_recv, (int)in, a.nparticles, a.x_off, a.y_off,
                                             g.d_free_slots, g.free_count, reuse, g.stream));
  g.free_count -= reuse;
  return grow;
}

}  // namespace neutral_abi