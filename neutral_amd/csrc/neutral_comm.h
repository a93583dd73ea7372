/*
 * neutral_comm.h -- what the C-ABI needs from the device side of the rank layer
 * (neutral_comm.hip).
 */
#ifndef NEUTRAL_AMD_COMM_H
#define NEUTRAL_AMD_COMM_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace neutral {

/* 1 / 0 until neutral_hip_comm_start() has run */
int comm_nranks();
int comm_rank();
int comm_transport();
/* in-place sum over the ranks of n 8-byte words in device memory (f64 or u64), on
 * `stream`: enqueued (RCCL) or completed on return (staged through the host) */
void comm_allreduce_sum(void* d_buf, size_t n, bool is_f64, hipStream_t stream);

/* personalised exchange of bytes in device memory: rank s hands matrix[s * n + d] bytes
 * to rank d (send buffer ordered by d, receive buffer by s); complete on return */
void comm_exchange_bytes(const void* d_send, void* d_recv, const uint64_t* matrix,
                         hipStream_t stream);

}  // namespace neutral
#endif
