// RCCL point-to-point transport for the slab halo exchange, loaded at run time (dlopen), so that libsph_amd.so has no
// link-time dependency on librccl and single-GPU users never load it.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sph {

constexpr size_t kRcclIdBytes = 128;  // sizeof(ncclUniqueId)

// All return 0 on success; on failure a message is written to err (errlen bytes).
int rccl_unique_id(void* id_out, char* err, size_t errlen);
int rccl_comm_create(void** comm_out, const void* id, int rank, int nranks, char* err, size_t errlen);
void rccl_comm_destroy(void* comm);
// One grouped exchange with the two neighbours on `stream`: floats of fp32 sent to and received from rank-1 and
// rank+1.  A negative peer rank means "no such neighbour".  Counts must agree pairwise between the ranks.
int rccl_exchange(void* comm, hipStream_t stream, int left, const float* send_left, size_t n_send_left,
                  float* recv_left, size_t n_recv_left, int right, const float* send_right, size_t n_send_right,
                  float* recv_right, size_t n_recv_right, char* err, size_t errlen);

}  // namespace sph
