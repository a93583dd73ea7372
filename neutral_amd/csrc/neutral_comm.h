/*
 * neutral_comm.h -- what the C-ABI needs from the device side of the rank layer
 * (neutral_comm.hip).
 */
#ifndef NEUTRAL_AMD_COMM_H
#define NEUTRAL_AMD_COMM_H

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace neutral {

/* 1 / 0 until neutral_hip_comm_start() has run */
int comm_nranks();
int comm_rank();
int comm_transport();
/* in-place sum over the ranks of n 8-byte words in device memory (f64 or u64), on
 * `stream`: enqueued (RCCL) or completed on return (staged through the host) */
void comm_allreduce_sum(void* d_buf, size_t n, bool is_f64, hipStream_t stream);

}  // namespace neutral
#endif
