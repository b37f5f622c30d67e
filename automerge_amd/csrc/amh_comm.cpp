#include "amh_comm.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>

// the few entry points of rccl.h this file uses, bound at run time (same ABI as NCCL 2.x)
namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclUint8 = 1, ncclUint64 = 5 };
struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* err = nullptr;
};
Api& api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, []() {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) { a.err = "librccl.so.1 not found"; return; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); ok = ok && p != nullptr; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) a.err = "librccl lacks an expected entry point";
  });
  return a;
}
}  // namespace

struct AmhComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

const char* amh_comm_make_id(uint8_t* id128) {
  Api& a = api();
  if (a.err) return a.err;
  ncclUniqueId id;
  ncclResult_t rc = a.GetUniqueId(&id);
  if (rc != 0) return a.GetErrorString(rc);
  memcpy(id128, id.internal, 128);
  return nullptr;
}
AmhComm* amh_comm_open(const uint8_t* id128, int rank, int world, int device, const char** err) {
  Api& a = api();
  if (a.err) { *err = a.err; return nullptr; }
  if (world < 1 || rank < 0 || rank >= world) { *err = "rank / world out of range"; return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; return nullptr; }
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  AmhComm* c = new AmhComm();
  c->rank = rank; c->world = world; c->device = device;
  ncclResult_t rc = a.CommInitRank(&c->comm, world, id, rank);
  if (rc != 0) { *err = a.GetErrorString(rc); delete c; return nullptr; }
  return c;
}
void amh_comm_close(AmhComm* c) {
  if (!c) return;
  if (c->comm) api().CommDestroy(c->comm);
  delete c;
}
int amh_comm_rank(const AmhComm* c) { return c->rank; }
int amh_comm_world(const AmhComm* c) { return c->world; }
const char* amh_comm_allgather_u64(AmhComm* c, const uint64_t* d_mine, uint64_t* d_all, void* stream) {
  Api& a = api();
  ncclResult_t rc = a.AllGather(d_mine, d_all, 1, ncclUint64, c->comm, (hipStream_t)stream);
  return rc ? a.GetErrorString(rc) : nullptr;
}
// One broadcast per contributing rank, grouped into a single launch: every rank knows all sizes, so all ranks issue
// the same sequence.  A ring all-gather over 7 xGMI links moves (world - 1) / world of the total per GPU either way;
// padding every contribution to the largest (the all_gather of torch.distributed) would send more.
const char* amh_comm_allgatherv(AmhComm* c, const uint8_t* d_mine, uint8_t* d_all, const uint64_t* sizes, const uint64_t* offs, void* stream) {
  Api& a = api();
  ncclResult_t rc = a.GroupStart();
  if (rc) return a.GetErrorString(rc);
  for (int r = 0; r < c->world && !rc; r++) {
    if (sizes[r] == 0) continue;
    rc = a.Broadcast(r == c->rank ? (const void*)d_mine : (const void*)(d_all + offs[r]), d_all + offs[r], (size_t)sizes[r], ncclUint8, r, c->comm, (hipStream_t)stream);
  }
  ncclResult_t rc2 = a.GroupEnd();
  if (rc) return a.GetErrorString(rc);
  return rc2 ? a.GetErrorString(rc2) : nullptr;
}
