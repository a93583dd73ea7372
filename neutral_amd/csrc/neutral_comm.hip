/*
 * neutral_comm.hip -- the device side of the rank layer: the end-of-step exchange
 * of the tally mesh between the GPUs of one node.
 *
 * Particles are sharded over the ranks (contiguous id ranges, host/comms_ranks.c),
 * the mesh is replicated; what the ranks owe each other per timestep is ONE
 * all-reduce (sum, f64, nx*ny elements: 1.28 MB at 400^2) of that step's tally
 * contributions and one of the four event counters.  The reference has no working
 * counterpart (its MPI blocks are compiled out, neutral_data.h:10-14); the hooks it
 * does call -- barrier() (main.c:75,112) and reduce_all_sum (omp3/neutral.c:530) --
 * are served by the same layer.
 *
 * Transport: RCCL over xGMI (ncclAllReduce on the stream the kernels run on, in
 * place, no host round trip).  librccl is loaded at run time (dlopen) so that the
 * library still loads where it is absent, and the communicator is brought up on a
 * helper thread with a time limit: if RCCL cannot be loaded, fails or does not come
 * up in time (or NEUTRAL_HIP_COMM=host asks for it: ranks that share one GPU in
 * tests), the exchange is staged through the host over the TCP links of
 * comms_ranks.c instead -- slower, same sums in a fixed rank order.
 */
#include "../../include/neutral_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../host/comms.h"
#include "neutral_comm.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  /* point to point, for the particle exchange of the decomposed-mesh mode (optional) */
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};

struct Comm {
  bool started = false;
  int transport = NEUTRAL_HIP_COMM_NONE;
  Rccl rccl;
  ncclComm_t comm = nullptr;
  int device = 0;
  bool first_collective_done = false;
  /* host staging buffer of the fallback transport */
  void* staging = nullptr;
  size_t staging_bytes = 0;
};

Comm c;

bool load_rccl(Rccl& r) {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) {
      break;
    }
  }
  if (!r.handle) {
    return false;
  }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
  r.Send = (decltype(r.Send))dlsym(r.handle, "ncclSend");
  r.Recv = (decltype(r.Recv))dlsym(r.handle, "ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
  return r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.GetErrorString;
}

/* The helper thread owns its job record, the communicator slot included: only the
 * waiting thread publishes job->comm to c.comm, and only after the helper said it is
 * done.  A helper that outlives its time limit keeps the record (it is never freed and
 * its communicator never destroyed: it may still be half-way through the bring-up). */
struct InitJob {
  ncclUniqueId id;
  int rank;
  int nranks;
  int device;
  ncclComm_t comm;
  ncclResult_t result;
  volatile int done;
};

void* init_thread(void* arg) {
  InitJob* j = (InitJob*)arg;
  (void)hipSetDevice(j->device);
  j->result = c.rccl.CommInitRank(&j->comm, j->nranks, j->id, j->rank);
  __atomic_store_n(&j->done, 1, __ATOMIC_RELEASE);
  return nullptr;
}

/* every rank reports whether its communicator came up; RCCL is used only if all did */
bool everybody(bool ok) {
  uint64_t v = ok ? 1u : 0u;
  comms_allreduce_u64(&v, 1, COMMS_MIN);
  return v != 0;
}

void* staging(size_t bytes) {
  if (bytes > c.staging_bytes) {
    if (c.staging) {
      (void)hipHostFree(c.staging);
    }
    if (hipHostMalloc(&c.staging, bytes, hipHostMallocDefault) != hipSuccess) {
      fprintf(stderr, "libneutral_hip: no pinned memory for the staged tally exchange.\n");
      exit(EXIT_FAILURE);
    }
    c.staging_bytes = bytes;
  }
  return c.staging;
}

void hip_or_die(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    fprintf(stderr, "libneutral_hip: %s failed: %s\n", what, hipGetErrorString(e));
    exit(EXIT_FAILURE);
  }
}

}  // namespace

namespace neutral {

int comm_nranks() { return c.started ? comms_nranks() : 1; }
int comm_rank() { return c.started ? comms_rank() : 0; }
int comm_transport() { return c.transport; }

void comm_allreduce_sum(void* d_buf, size_t n, bool is_f64, hipStream_t stream) {
  if (comm_nranks() == 1 || n == 0) {
    return;
  }
  if (c.transport == NEUTRAL_HIP_COMM_RCCL) {
    const ncclResult_t r = c.rccl.AllReduce(d_buf, d_buf, n, is_f64 ? ncclDouble : ncclUint64,
                                            ncclSum, c.comm, stream);
    if (r != ncclSuccess) {
      fprintf(stderr, "libneutral_hip: rank %d: ncclAllReduce failed: %s\n", comm_rank(),
              c.rccl.GetErrorString(r));
      exit(EXIT_FAILURE); /* a rank that leaves takes the job down: no silent partial sums */
    }
    if (!c.first_collective_done) {
      /* the first collective of a communicator is where a broken fabric shows: wait for
       * it here, with a limit, and say so instead of hanging in some later wait */
      const int limit_s = getenv("NEUTRAL_COMM_TIMEOUT") ? atoi(getenv("NEUTRAL_COMM_TIMEOUT")) : 120;
      struct timespec t0, t;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (;;) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) {
          break;
        }
        if (q != hipErrorNotReady) {
          hip_or_die(q, "the first RCCL all-reduce");
        }
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((t.tv_sec - t0.tv_sec) > limit_s) {
          fprintf(stderr, "libneutral_hip: rank %d: the first RCCL all-reduce did not finish within "
                          "%d s; set NEUTRAL_HIP_COMM=host to stage the exchange through the "
                          "hosts.\n", comm_rank(), limit_s);
          _exit(EXIT_FAILURE);
        }
        struct timespec nap = {0, 200000};
        nanosleep(&nap, nullptr);
      }
      c.first_collective_done = true;
    }
    return;
  }
  /* staged through the host: device -> pinned buffer -> TCP star -> device */
  void* h = staging(n * 8);
  hip_or_die(hipMemcpyAsync(h, d_buf, n * 8, hipMemcpyDeviceToHost, stream), "staging D2H");
  hip_or_die(hipStreamSynchronize(stream), "staging sync");
  if (is_f64) {
    comms_allreduce_f64((double*)h, n, COMMS_SUM);
  } else {
    comms_allreduce_u64((uint64_t*)h, n, COMMS_SUM);
  }
  hip_or_die(hipMemcpyAsync(d_buf, h, n * 8, hipMemcpyHostToDevice, stream), "staging H2D");
  hip_or_die(hipStreamSynchronize(stream), "staging sync");
}

/* Personalised exchange of bytes in device memory (the particles that cross between
 * the ranks' blocks of a decomposed mesh): rank s hands matrix[s * n + d] bytes to rank
 * d, send buffer ordered by d, receive buffer ordered by s.  ncclSend / ncclRecv in one
 * group over xGMI when RCCL is up, staged through the host otherwise.  Complete on
 * return. */
void comm_exchange_bytes(const void* d_send, void* d_recv, const uint64_t* matrix,
                         hipStream_t stream) {
  const int n = comm_nranks();
  const int me = comm_rank();
  size_t out = 0, in = 0;
  for (int d = 0; d < n; ++d) out += (size_t)matrix[(size_t)me * n + d];
  for (int s2 = 0; s2 < n; ++s2) in += (size_t)matrix[(size_t)s2 * n + me];
  if (c.transport == NEUTRAL_HIP_COMM_RCCL && c.rccl.Send && c.rccl.Recv && c.rccl.GroupStart &&
      c.rccl.GroupEnd) {
    ncclResult_t r = c.rccl.GroupStart();
    size_t so = 0, ro = 0;
    for (int peer = 0; peer < n && r == ncclSuccess; ++peer) {
      const size_t sb = (size_t)matrix[(size_t)me * n + peer];
      const size_t rb = (size_t)matrix[(size_t)peer * n + me];
      if (sb) r = c.rccl.Send((const char*)d_send + so, sb, ncclChar, peer, c.comm, stream);
      if (rb && r == ncclSuccess) {
        r = c.rccl.Recv((char*)d_recv + ro, rb, ncclChar, peer, c.comm, stream);
      }
      so += sb;
      ro += rb;
    }
    if (r == ncclSuccess) r = c.rccl.GroupEnd();
    if (r != ncclSuccess) {
      fprintf(stderr, "libneutral_hip: rank %d: RCCL send/recv failed: %s\n", me,
              c.rccl.GetErrorString(r));
      exit(EXIT_FAILURE);
    }
    hip_or_die(hipStreamSynchronize(stream), "particle exchange");
    return;
  }
  char* h = (char*)staging(out + in + 16);
  if (out) hip_or_die(hipMemcpyAsync(h, d_send, out, hipMemcpyDeviceToHost, stream), "D2H");
  hip_or_die(hipStreamSynchronize(stream), "particle exchange");
  comms_alltoallv(h, h + out, matrix);
  if (in) hip_or_die(hipMemcpyAsync(d_recv, h + out, in, hipMemcpyHostToDevice, stream), "H2D");
  hip_or_die(hipStreamSynchronize(stream), "particle exchange");
}

}  // namespace neutral

extern "C" {

int neutral_hip_comm_start(void) {
  if (c.started) {
    return c.transport;
  }
  comms_start_from_env();
  c.started = true;
  const int nranks = comms_nranks();
  const int rank = comms_rank();
  if (nranks == 1 && !getenv("NEUTRAL_HIP_COMM")) {
    c.transport = NEUTRAL_HIP_COMM_NONE;
    return c.transport;
  }
  (void)hipGetDevice(&c.device);
  const char* want = getenv("NEUTRAL_HIP_COMM");
  bool try_rccl = !(want && strcmp(want, "host") == 0);
  bool ok = false;
  if (try_rccl) {
    ok = load_rccl(c.rccl);
    if (!ok && rank == 0) {
      fprintf(stderr, "libneutral_hip: librccl could not be loaded (%s); the tally exchange "
                      "is staged through the host.\n", dlerror());
    }
    /* the id comes from rank 0: everybody must take part in the broadcast, whether
     * its own load succeeded or not */
    /* (on the heap: a helper thread that outlives its time limit keeps using it) */
    InitJob* job = (InitJob*)calloc(1, sizeof(InitJob));
    if (!job) {
      fprintf(stderr, "libneutral_hip: out of memory.\n");
      exit(EXIT_FAILURE);
    }
    if (rank == 0 && ok) {
      ok = c.rccl.GetUniqueId(&job->id) == ncclSuccess;
    }
    comms_bcast_bytes(&job->id, sizeof(job->id));
    ok = everybody(ok);
    bool timed_out = false;
    if (ok) {
      job->rank = rank;
      job->nranks = nranks;
      job->device = c.device;
      pthread_t th;
      const int limit_s = getenv("NEUTRAL_COMM_TIMEOUT") ? atoi(getenv("NEUTRAL_COMM_TIMEOUT")) : 120;
      if (pthread_create(&th, nullptr, init_thread, job) != 0) {
        ok = false;
        free(job);
      } else {
        struct timespec t0, t;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
          if (__atomic_load_n(&job->done, __ATOMIC_ACQUIRE)) {
            pthread_join(th, nullptr);
            ok = job->result == ncclSuccess;
            if (ok) {
              c.comm = job->comm; /* published here, by the thread that uses it */
            } else {
              fprintf(stderr, "libneutral_hip: rank %d: ncclCommInitRank failed: %s\n", rank,
                      c.rccl.GetErrorString(job->result));
            }
            free(job);
            break;
          }
          clock_gettime(CLOCK_MONOTONIC, &t);
          if ((t.tv_sec - t0.tv_sec) > limit_s) {
            /* the thread is left behind with its job record (it cannot be cancelled
             * safely, and nothing of the record is touched again from here) */
            fprintf(stderr, "libneutral_hip: rank %d: RCCL did not come up within %d s.\n", rank,
                    limit_s);
            pthread_detach(th);
            ok = false;
            timed_out = true;
            break;
          }
          struct timespec nap = {0, 2000000};
          nanosleep(&nap, nullptr);
        }
      }
      /* A bring-up that FAILED leaves nothing behind: every rank falls back to the host
       * route.  One that TIMED OUT leaves a thread inside RCCL on this rank and, on the
       * ranks where it succeeded, a communicator joined by a peer that gave up: the job
       * ends here, on every rank, instead of running on next to that. */
      if (!everybody(!timed_out)) {
        fprintf(stderr, "libneutral_hip: rank %d: the RCCL bring-up timed out on some rank; "
                        "set NEUTRAL_HIP_COMM=host to stage the exchange through the hosts, or "
                        "raise NEUTRAL_COMM_TIMEOUT.\n", rank);
        _exit(EXIT_FAILURE);
      }
      ok = everybody(ok);
    } else {
      free(job);
    }
  }
  c.transport = ok ? NEUTRAL_HIP_COMM_RCCL : NEUTRAL_HIP_COMM_HOST;
  if (rank == 0 && !getenv("NEUTRAL_HIP_QUIET")) {
    fprintf(stderr, "libneutral_hip: %d ranks, tally exchange over %s.\n", nranks,
            ok ? "RCCL" : "the host (TCP)");
  }
  return c.transport;
}

void neutral_hip_comm_stop(void) {
  if (c.comm) {
    (void)c.rccl.CommDestroy(c.comm);
    c.comm = nullptr;
  }
  c.first_collective_done = false;
  if (c.staging) {
    (void)hipHostFree(c.staging);
    c.staging = nullptr;
    c.staging_bytes = 0;
  }
  c.transport = NEUTRAL_HIP_COMM_NONE;
  c.started = false;
  comms_stop();
}

int neutral_hip_comm_rank(void) { return neutral::comm_rank(); }
int neutral_hip_comm_nranks(void) { return neutral::comm_nranks(); }
int neutral_hip_comm_transport(void) { return c.transport; }

/* RCCL's own version number (ncclGetVersion: e.g. 22203 for 2.22.3), 0 when librccl cannot be
 * loaded here -- what a first multi-GPU record should say it ran on */
int neutral_hip_comm_rccl_version(void) {
  if (!c.rccl.handle && !load_rccl(c.rccl)) {
    return 0;
  }
  typedef ncclResult_t (*GetVersion)(int*);
  GetVersion get = (GetVersion)dlsym(c.rccl.handle, "ncclGetVersion");
  int v = 0;
  if (!get || get(&v) != 0) {
    return 0;
  }
  return v;
}

void neutral_hip_comm_allreduce_f64(double* device_buf, size_t n, void* hip_stream) {
  neutral::comm_allreduce_sum(device_buf, n, true, (hipStream_t)hip_stream);
}

double neutral_hip_comm_max(double v) {
  if (c.started) {
    comms_allreduce_f64(&v, 1, COMMS_MAX);
  }
  return v;
}

void neutral_hip_comm_barrier(void) {
  if (c.started) {
    comms_barrier();
  }
}

/* used by the host layer (host.c: initialise_devices, barrier) */
void neutral_hip_bind_rank_device(int local_rank) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fprintf(stderr, "libneutral_hip: no GPU visible to rank %d.\n", comms_rank());
    exit(EXIT_FAILURE);
  }
  const char* share = getenv("NEUTRAL_HIP_SHARE_DEVICE"); /* tests: every rank on GPU 0 */
  const int dev = (share && atoi(share)) ? 0 : local_rank;
  if (dev >= ndev) {
    fprintf(stderr, "libneutral_hip: rank %d wants GPU %d but only %d are visible.\n",
            comms_rank(), dev, ndev);
    exit(EXIT_FAILURE);
  }
  hip_or_die(hipSetDevice(dev), "hipSetDevice");
  (void)neutral_hip_comm_start();
}

void neutral_hip_comm_barrier_device(void) { hip_or_die(hipDeviceSynchronize(), "barrier"); }

/* One-rank check that the RCCL route works on this machine: loads librccl, brings up
 * a communicator of one rank on the current device and all-reduces n doubles in
 * place (the identity).  Returns 0 on success, 1 if RCCL cannot be loaded, 2 on any
 * RCCL or HIP failure. */
int neutral_hip_comm_selftest(int n) {
  Rccl r;
  if (!load_rccl(r)) {
    return 1;
  }
  ncclUniqueId id;
  ncclComm_t comm = nullptr;
  if (r.GetUniqueId(&id) != ncclSuccess || r.CommInitRank(&comm, 1, id, 0) != ncclSuccess) {
    return 2;
  }
  double* d = nullptr;
  double* h = (double*)malloc(sizeof(double) * (size_t)n);
  int rc = 0;
  if (!h || hipMalloc((void**)&d, sizeof(double) * (size_t)n) != hipSuccess) {
    rc = 2;
  } else {
    for (int i = 0; i < n; ++i) h[i] = 0.5 * i;
    if (hipMemcpy(d, h, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
        r.AllReduce(d, d, (size_t)n, ncclDouble, ncclSum, comm, nullptr) != ncclSuccess ||
        hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(h, d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = 2;
    } else {
      for (int i = 0; i < n; ++i) {
        if (h[i] != 0.5 * i) rc = 2;
      }
    }
  }
  if (d) (void)hipFree(d);
  free(h);
  (void)r.CommDestroy(comm);
  return rc;
}

}  // extern "C"
