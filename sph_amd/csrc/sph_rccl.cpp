// See sph_rccl.hpp.  Only the handful of RCCL entry points the slab exchange needs are resolved.
#include "sph_rccl.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

namespace sph {

namespace {

struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Api g_api;
std::mutex g_mu;

template <class F>
bool resolve(void* lib, const char* name, F* fn) {
  *fn = reinterpret_cast<F>(dlsym(lib, name));
  return *fn != nullptr;
}

int load(char* err, size_t errlen) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_api.lib) return 0;
  // a process that already runs RCCL (torch.distributed's backend) gets that copy: same soname
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names)
    if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!lib) {
    snprintf(err, errlen, "cannot load librccl: %s", dlerror());
    return 1;
  }
  Api a;
  a.lib = lib;
  if (!resolve(lib, "ncclGetUniqueId", &a.GetUniqueId) || !resolve(lib, "ncclCommInitRank", &a.CommInitRank) ||
      !resolve(lib, "ncclCommDestroy", &a.CommDestroy) || !resolve(lib, "ncclGroupStart", &a.GroupStart) ||
      !resolve(lib, "ncclGroupEnd", &a.GroupEnd) || !resolve(lib, "ncclSend", &a.Send) ||
      !resolve(lib, "ncclRecv", &a.Recv) || !resolve(lib, "ncclGetErrorString", &a.GetErrorString)) {
    snprintf(err, errlen, "librccl lacks a required entry point");
    dlclose(lib);
    return 1;
  }
  g_api = a;
  return 0;
}

int check(ncclResult_t r, const char* what, char* err, size_t errlen) {
  if (r == ncclSuccess) return 0;
  snprintf(err, errlen, "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "RCCL error");
  return 1;
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == kRcclIdBytes, "ncclUniqueId size");

int rccl_unique_id(void* id_out, char* err, size_t errlen) {
  if (load(err, errlen)) return 1;
  ncclUniqueId id;
  if (check(g_api.GetUniqueId(&id), "ncclGetUniqueId", err, errlen)) return 1;
  memcpy(id_out, &id, sizeof id);
  return 0;
}

int rccl_comm_create(void** comm_out, const void* id_in, int rank, int nranks, char* err, size_t errlen) {
  if (load(err, errlen)) return 1;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof id);
  ncclComm_t comm = nullptr;
  if (check(g_api.CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank", err, errlen)) return 1;
  *comm_out = comm;
  return 0;
}

void rccl_comm_destroy(void* comm) {
  if (comm && g_api.CommDestroy) (void)g_api.CommDestroy((ncclComm_t)comm);
}

int rccl_exchange(void* comm_, hipStream_t stream, int left, const float* send_left, size_t n_send_left,
                  float* recv_left, size_t n_recv_left, int right, const float* send_right, size_t n_send_right,
                  float* recv_right, size_t n_recv_right, char* err, size_t errlen) {
  ncclComm_t comm = (ncclComm_t)comm_;
  if (check(g_api.GroupStart(), "ncclGroupStart", err, errlen)) return 1;
  int bad = 0;
  if (left >= 0) {
    if (n_send_left) bad |= check(g_api.Send(send_left, n_send_left, ncclFloat, left, comm, stream), "ncclSend", err, errlen);
    if (n_recv_left) bad |= check(g_api.Recv(recv_left, n_recv_left, ncclFloat, left, comm, stream), "ncclRecv", err, errlen);
  }
  if (right >= 0) {
    if (n_send_right) bad |= check(g_api.Send(send_right, n_send_right, ncclFloat, right, comm, stream), "ncclSend", err, errlen);
    if (n_recv_right) bad |= check(g_api.Recv(recv_right, n_recv_right, ncclFloat, right, comm, stream), "ncclRecv", err, errlen);
  }
  if (check(g_api.GroupEnd(), "ncclGroupEnd", err, errlen)) return 1;
  return bad;
}

}  // namespace sph
