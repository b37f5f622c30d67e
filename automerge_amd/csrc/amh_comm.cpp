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
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* err = nullptr;
};
Api& api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, []() {
    // An RCCL that is already mapped into the process (PyTorch ships its own librccl.so and loads it with torch) is the
    // one to use: a second copy next to it (ROCm's librccl.so.1) runs its own static destructors over the same HIP
    // runtime state at exit ("double free or corruption" after the last test).  So: probe with RTLD_NOLOAD under both
    // names first, load only when none is there, and keep the library's symbols out of the global namespace.
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (a.lib) break;
    }
    if (!a.lib)
      for (const char* name : {"librccl.so.1", "librccl.so"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a.lib) break;
      }
    if (!a.lib) { a.err = "librccl.so.1 not found"; return; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); ok = ok && p != nullptr; return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
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
// All-gather(v) as a direct all-pairs exchange: inside one group every rank sends its records to each peer and receives
// each peer's records at their place (every rank knows all sizes, so all ranks issue matching pairs).  xGMI is a full
// mesh of point-to-point links, so each link carries exactly one shard in each direction; a ring, or
// one broadcast per rank, would pass every shard over up to world - 1 hops.  The rank's own records are a device copy.
const char* amh_comm_allgatherv(AmhComm* c, const uint8_t* d_mine, uint8_t* d_all, const uint64_t* sizes, const uint64_t* offs, void* stream) {
  Api& a = api();
  hipStream_t s = (hipStream_t)stream;
  const uint64_t mine = sizes[c->rank];
  if (mine && d_mine != d_all + offs[c->rank] &&
      hipMemcpyAsync(d_all + offs[c->rank], d_mine, (size_t)mine, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return "hipMemcpyAsync failed";
  if (c->world == 1) return nullptr;
  ncclResult_t rc = a.GroupStart();
  if (rc) return a.GetErrorString(rc);
  for (int r = 0; r < c->world && !rc; r++) {
    if (r == c->rank) continue;
    if (mine) rc = a.Send(d_mine, (size_t)mine, ncclUint8, r, c->comm, s);
    if (!rc && sizes[r]) rc = a.Recv(d_all + offs[r], (size_t)sizes[r], ncclUint8, r, c->comm, s);
  }
  ncclResult_t rc2 = a.GroupEnd();
  if (rc) return a.GetErrorString(rc);
  return rc2 ? a.GetErrorString(rc2) : nullptr;
}
