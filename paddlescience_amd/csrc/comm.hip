// comm.hip -- the gradient all-reduce of the data-parallel step in the C ABI (ppsci_comm_init /
// ppsci_allreduce_grads / ppsci_allgather; the reference: `paddle.distributed` fused all-reduce of the parameter
// gradients, /root/reference/ppsci/solver/train.py:168-171, solver.py:1005-1044).
//
// RCCL is NOT linked into libppsci_hip.so: torch carries its own copy of librccl, and a second one with the same soname
// pulled in at load time would decide which of the two the whole process uses.  The library is resolved lazily, at
// ppsci_comm_init(): an already loaded librccl (torch's, once torch.distributed's nccl backend has been used) is
// preferred (RTLD_NOLOAD), the system one under /opt/rocm/lib is the fallback.  One communicator per process (one
// process per GPU).  The Python host uses this path when PPSCI_NATIVE_ALLREDUCE=1 (engine.Engine.allreduce); the
// default remains torch.distributed, which is the same RCCL underneath.
#include "ppsci_common.h"

#include <stdint.h>
#include <string.h>

extern "C" void ppsci_set_error(const char* fmt, ...);

#ifdef PPSCI_EMU
// the CPU emulator has no device collectives: every entry point reports it (the gloo tests cover the Python path)
extern "C" int ppsci_comm_unique_id(void*) { ppsci_set_error("comm: not available in the emulator build"); return PPSCI_E_UNSUPPORTED; }
extern "C" int ppsci_comm_init(int, int, const void*) { ppsci_set_error("comm: not available in the emulator build"); return PPSCI_E_UNSUPPORTED; }
extern "C" int ppsci_comm_world_size(void) { return 0; }
extern "C" int ppsci_allreduce_sum(float*, int64_t, void*) { ppsci_set_error("comm: not available in the emulator build"); return PPSCI_E_UNSUPPORTED; }
extern "C" int ppsci_allgather(const float*, float*, int64_t, void*) { ppsci_set_error("comm: not available in the emulator build"); return PPSCI_E_UNSUPPORTED; }
extern "C" int ppsci_comm_destroy(void) { return PPSCI_OK; }
#else
#include <dlfcn.h>
#include <hip/hip_runtime.h>

namespace {
// the part of rccl.h that is used (RCCL 2.x ABI: /opt/rocm/include/rccl/rccl.h:40-43, :187, :220, :260, :339, :448, :466, :611, :678)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int /*dtype*/, int /*op*/, Comm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int /*dtype*/, Comm, hipStream_t);
const int kFloat32 = 7, kSum = 0;

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  AllReduceFn all_reduce = nullptr;
  AllGatherFn all_gather = nullptr;
  Comm comm = nullptr;
  int world = 0, rank = -1;
} g;

bool load() {
  if (g.handle) return true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // the copy the process already uses, if any
    if (g.handle) break;
  }
  for (int i = 0; !g.handle && i < 3; ++i) g.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!g.handle) {
    ppsci_set_error("comm: cannot load librccl (%s)", dlerror());
    return false;
  }
  g.get_unique_id = (GetUniqueIdFn)dlsym(g.handle, "ncclGetUniqueId");
  g.comm_init_rank = (CommInitRankFn)dlsym(g.handle, "ncclCommInitRank");
  g.comm_destroy = (CommDestroyFn)dlsym(g.handle, "ncclCommDestroy");
  g.error_string = (GetErrorStringFn)dlsym(g.handle, "ncclGetErrorString");
  g.all_reduce = (AllReduceFn)dlsym(g.handle, "ncclAllReduce");
  g.all_gather = (AllGatherFn)dlsym(g.handle, "ncclAllGather");
  if (!g.get_unique_id || !g.comm_init_rank || !g.comm_destroy || !g.all_reduce || !g.all_gather) {
    ppsci_set_error("comm: librccl lacks an expected symbol");
    g.handle = nullptr;
    return false;
  }
  return true;
}

int fail(const char* what, int rc) {
  ppsci_set_error("comm: %s failed: %s (%d)", what, g.error_string ? g.error_string(rc) : "?", rc);
  return PPSCI_E_LAUNCH;
}
}  // namespace

// 128 bytes; rank 0 calls it and ships the bytes to the other ranks by any host-side channel
extern "C" int ppsci_comm_unique_id(void* out128) {
  if (!out128) { ppsci_set_error("comm_unique_id: null argument"); return PPSCI_E_INVALID; }
  if (!load()) return PPSCI_E_UNSUPPORTED;
  UniqueId id;
  const int rc = g.get_unique_id(&id);
  if (rc != 0) return fail("ncclGetUniqueId", rc);
  memcpy(out128, &id, sizeof(id));
  return PPSCI_OK;
}

// collective over all ranks; the calling thread's current HIP device is the rank's GPU
extern "C" int ppsci_comm_init(int rank, int world, const void* id128) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) { ppsci_set_error("comm_init: invalid argument"); return PPSCI_E_INVALID; }
  if (!load()) return PPSCI_E_UNSUPPORTED;
  if (g.comm) { ppsci_set_error("comm_init: a communicator exists already (one per process)"); return PPSCI_E_INVALID; }
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  const int rc = g.comm_init_rank(&g.comm, world, id, rank);
  if (rc != 0) { g.comm = nullptr; return fail("ncclCommInitRank", rc); }
  g.world = world;
  g.rank = rank;
  return PPSCI_OK;
}

extern "C" int ppsci_comm_world_size(void) { return g.comm ? g.world : 0; }

// buf <- sum over ranks of buf (in place), ordered on `stream`: train.py:168-171 on the ONE flat gradient buffer
extern "C" int ppsci_allreduce_sum(float* buf, int64_t n, void* stream) {
  if (!g.comm) { ppsci_set_error("allreduce_sum: ppsci_comm_init has not been called"); return PPSCI_E_INVALID; }
  if (!buf || n < 1) { ppsci_set_error("allreduce_sum: invalid argument"); return PPSCI_E_INVALID; }
  const int rc = g.all_reduce(buf, buf, (size_t)n, kFloat32, kSum, g.comm, (hipStream_t)stream);
  return rc == 0 ? PPSCI_OK : fail("ncclAllReduce", rc);
}

// recv[r*n : (r+1)*n] <- rank r's send[0:n]  (evaluation gather, /root/reference/ppsci/utils/misc.py all_gather)
extern "C" int ppsci_allgather(const float* send, float* recv, int64_t n, void* stream) {
  if (!g.comm) { ppsci_set_error("allgather: ppsci_comm_init has not been called"); return PPSCI_E_INVALID; }
  if (!send || !recv || n < 1) { ppsci_set_error("allgather: invalid argument"); return PPSCI_E_INVALID; }
  const int rc = g.all_gather(send, recv, (size_t)n, kFloat32, g.comm, (hipStream_t)stream);
  return rc == 0 ? PPSCI_OK : fail("ncclAllGather", rc);
}

extern "C" int ppsci_comm_destroy(void) {
  if (g.comm) {
    const int rc = g.comm_destroy(g.comm);
    g.comm = nullptr;
    g.world = 0;
    if (rc != 0) return fail("ncclCommDestroy", rc);
  }
  return PPSCI_OK;
}
#endif
