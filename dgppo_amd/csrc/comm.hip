// C1: the gradient exchange of the data-parallel update — one in-place all-reduce(sum) of the
// flat [g_policy | g_Vl | g_Vh | scalars] fp32 buffer per minibatch over RCCL (xGMI inside a node), on the caller's
// stream.  The reference has no counterpart (it is single-device); this is what jax.lax.pmean of the three
// gradient trees would be in a pmap'ed update_inner (dgppo/algo/dgppo.py:188-294).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy the process has already loaded if any) so that
// libdgppo_hip.so itself loads on machines without RCCL; every entry point fails loudly when it is missing.
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

namespace {

typedef struct { char internal[DGPPO_COMM_ID_BYTES]; } rccl_unique_id;   // ncclUniqueId: 128 opaque bytes
typedef void* rccl_comm_t;
enum { RCCL_SUCCESS = 0, RCCL_FLOAT32 = 7, RCCL_SUM = 0 };               // ncclSuccess, ncclFloat32, ncclSum (rccl.h)

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;                 // optional (diagnostics only)
};

RcclApi g_api;

bool load_rccl() {
  if (g_api.handle) return true;
  const char* names[] = {getenv("DGPPO_RCCL_LIB"), "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);          // the copy already in the process (PyTorch ships one)
    if (!h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    dgppo_set_error("RCCL not found (tried $DGPPO_RCCL_LIB, librccl.so.1, librccl.so): %s", dlerror());
    return false;
  }
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  a.GetVersion = (decltype(a.GetVersion))dlsym(h, "ncclGetVersion");
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.GetErrorString) {
    dgppo_set_error("RCCL library lacks one of ncclGetUniqueId/CommInitRank/AllReduce/CommDestroy/GetErrorString");
    return false;
  }
  g_api = a;
  return true;
}

struct Comm {
  uint32_t magic;
  rccl_comm_t comm;
  int rank, world;
};
constexpr uint32_t COMM_MAGIC = 0xD6CC0C01u;

int32_t rccl_fail(const char* what, int rc) {
  dgppo_set_error("%s failed: %s (ncclResult %d)", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?", rc);
  return 1000 + rc;   // > 0 like a hipError_t, offset so the two ranges cannot be confused
}

}  // namespace

extern "C" int32_t dgppo_comm_unique_id(uint8_t* id_out) {
  DGPPO_REQUIRE(id_out != nullptr, "comm_unique_id: id_out is NULL");
  if (!load_rccl()) return -1;
  rccl_unique_id id;
  const int rc = g_api.GetUniqueId(&id);
  if (rc != RCCL_SUCCESS) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, DGPPO_COMM_ID_BYTES);
  return 0;
}

extern "C" int32_t dgppo_comm_version(int32_t* version_out) {
  DGPPO_REQUIRE(version_out != nullptr, "comm_version: version_out is NULL");
  *version_out = 0;
  if (!load_rccl()) return -1;
  if (!g_api.GetVersion) { dgppo_set_error("RCCL library has no ncclGetVersion"); return -1; }
  int v = 0;
  const int rc = g_api.GetVersion(&v);
  if (rc != RCCL_SUCCESS) return rccl_fail("ncclGetVersion", rc);
  *version_out = v;
  return 0;
}

extern "C" int32_t dgppo_comm_init(const uint8_t* id, int32_t rank, int32_t world, void** comm_out) {
  DGPPO_REQUIRE(id != nullptr && comm_out != nullptr, "comm_init: NULL argument");
  DGPPO_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d outside world %d", rank, world);
  *comm_out = nullptr;
  if (!load_rccl()) return -1;
  rccl_unique_id uid;
  memcpy(uid.internal, id, DGPPO_COMM_ID_BYTES);
  rccl_comm_t c = nullptr;
  const int rc = g_api.CommInitRank(&c, world, uid, rank);   // binds to the calling thread's current HIP device
  if (rc != RCCL_SUCCESS) return rccl_fail("ncclCommInitRank", rc);
  Comm* h = (Comm*)malloc(sizeof(Comm));
  DGPPO_REQUIRE(h != nullptr, "comm_init: out of host memory");
  h->magic = COMM_MAGIC; h->comm = c; h->rank = rank; h->world = world;
  *comm_out = h;
  return 0;
}

extern "C" int32_t dgppo_comm_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream) {
  Comm* h = (Comm*)comm;
  DGPPO_REQUIRE(h != nullptr && h->magic == COMM_MAGIC, "comm_allreduce: not a communicator from dgppo_comm_init");
  DGPPO_REQUIRE(count >= 0, "comm_allreduce: count < 0");
  if (count == 0) return 0;
  DGPPO_REQUIRE(buf != nullptr, "comm_allreduce: buf is NULL");
  const int rc = g_api.AllReduce(buf, buf, (size_t)count, RCCL_FLOAT32, RCCL_SUM, h->comm, (hipStream_t)stream);
  if (rc != RCCL_SUCCESS) return rccl_fail("ncclAllReduce", rc);
  return 0;
}

extern "C" int32_t dgppo_comm_destroy(void* comm) {
  Comm* h = (Comm*)comm;
  if (h == nullptr) return 0;
  DGPPO_REQUIRE(h->magic == COMM_MAGIC, "comm_destroy: not a communicator from dgppo_comm_init");
  const int rc = g_api.CommDestroy(h->comm);
  h->magic = 0;
  free(h);
  if (rc != RCCL_SUCCESS) return rccl_fail("ncclCommDestroy", rc);
  return 0;
}
