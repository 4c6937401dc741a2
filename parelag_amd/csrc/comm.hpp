// RCCL neighbour exchange over xGMI: replaces hypre_ParCSRCommHandleCreate/Destroy (persistent MPI
// Isend/Irecv + Waitall; call site src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:189-193).
// librccl is resolved at run time (dlopen) and only when nranks > 1, so a single-GPU process never
// loads it.
#pragma once
#include "common.hpp"

namespace elag {

bool comm_active();
void comm_destroy();
void comm_unique_id(void *id128);
void comm_init(int rank, int nranks, const void *id128);

// One grouped send/recv round with the listed neighbours, enqueued on `st`.  Offsets are in elements
// of `elem_bytes` bytes.  sendbuf/recvbuf are device pointers.
void comm_neighbor_exchange(const void *sendbuf, int nsend, const int *send_procs, const int *send_starts,
                            void *recvbuf, int nrecv, const int *recv_procs, const int *recv_starts,
                            size_t elem_bytes, hipStream_t st);
// all-gather of variable-size double segments: recv[displs[r] .. +counts[r]) <- rank r's `send` (counts[rank] doubles)
void comm_allgatherv(const double *send, double *recv, const int *counts, const int *displs, hipStream_t st);
// 8-byte all-reduce of a host scalar (op 0 = sum, 1 = max); identity when nranks == 1.
void comm_allreduce_scalar(double *value, int op);
// all-reduce of a small device array in place on the compute stream
void comm_allreduce_device(double *d_values, int n, int op, hipStream_t st);

}  // namespace elag
