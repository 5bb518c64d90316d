// RCCL behind the C ABI: the sum of the per-rank partial node aggregates (SURVEY.md section 8b's "allreduce_sum
// wrapper over RCCL"; the north star's RCCL all-reduce of partial node aggregates over xGMI).  Host code only.
//
// librccl is resolved at first use with dlopen("librccl.so.1") -- no link-time dependency, so the library loads on a
// box without RCCL, and inside a PyTorch process the loader hands back the copy torch already mapped (same soname),
// not a second one.  One process per GPU; the caller creates the 128-byte unique id on one rank
// (lpg_comm_unique_id), ships it to the others by any host channel (the torch.distributed store, MPI, a file), and
// every rank calls lpg_comm_init.  Collectives only enqueue work on `stream`.
#include <dlfcn.h>
#include <string.h>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

struct UniqueId { char internal[128]; };            // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9;

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

char g_dl_error[256] = "no loader error";

Rccl load_rccl() {
  Rccl r;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
    if (const char* e = dlerror()) snprintf(g_dl_error, sizeof(g_dl_error), "%s", e);     // dlerror() clears itself: read once
  }
  if (!r.handle) return r;
  auto sym = [&](const char* s) { return dlsym(r.handle, s); };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(sym("ncclReduceScatter"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter;
  return r;
}

Rccl& rccl() {
  static Rccl r = load_rccl();        // function-local static: initialised once, thread-safe (C++11)
  return r;
}

int need_rccl() {
  if (rccl().ok) return LPG_OK;
  set_error("RCCL is not available: dlopen(librccl.so.1) failed or a symbol is missing (%s)", g_dl_error);
  return LPG_E_UNSUPPORTED;
}

int check_nccl(int rc, const char* what) {
  if (rc == 0) return LPG_OK;
  set_error("%s: RCCL error %d (%s)", what, rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
  return LPG_E_LAUNCH;
}

int dtype_of(int is_bf16) { return is_bf16 ? kNcclBfloat16 : kNcclFloat32; }
}  // namespace

LPG_EXPORT int lpg_comm_available(void) { return rccl().ok ? 1 : 0; }

LPG_EXPORT int lpg_comm_unique_id(void* id128) {
  LPG_REQUIRE(id128, "comm_unique_id: null pointer");
  if (int rc = need_rccl()) return rc;
  UniqueId id;
  if (int rc = check_nccl(rccl().GetUniqueId(&id), "ncclGetUniqueId")) return rc;
  memcpy(id128, &id, sizeof(id));
  return LPG_OK;
}

LPG_EXPORT int lpg_comm_init(void** comm, int world, int rank, const void* id128) {
  LPG_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
  if (int rc = need_rccl()) return rc;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  return check_nccl(rccl().CommInitRank(comm, world, id, rank), "ncclCommInitRank");     // binds the CURRENT hip device
}

LPG_EXPORT int lpg_comm_destroy(void* comm) {
  if (!comm) return LPG_OK;
  if (int rc = need_rccl()) return rc;
  return check_nccl(rccl().CommDestroy(comm), "ncclCommDestroy");
}

LPG_EXPORT int lpg_allreduce_sum(void* comm, const void* send, void* recv, int64_t count, int is_bf16, void* stream) {
  LPG_REQUIRE(comm && count >= 0 && (count == 0 || (send && recv)), "allreduce_sum: bad arguments");
  if (count == 0) return LPG_OK;
  if (int rc = need_rccl()) return rc;
  return check_nccl(rccl().AllReduce(send, recv, static_cast<size_t>(count), dtype_of(is_bf16), kNcclSum, comm, as_stream(stream)),
                    "ncclAllReduce");
}

LPG_EXPORT int lpg_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t recv_count, int is_bf16, void* stream) {
  LPG_REQUIRE(comm && recv_count >= 0 && (recv_count == 0 || (send && recv)), "reduce_scatter_sum: bad arguments");
  if (recv_count == 0) return LPG_OK;
  if (int rc = need_rccl()) return rc;
  return check_nccl(rccl().ReduceScatter(send, recv, static_cast<size_t>(recv_count), dtype_of(is_bf16), kNcclSum, comm,
                                         as_stream(stream)), "ncclReduceScatter");
}
