// Data-parallel gradient exchange over RCCL (xGMI), behind the C ABI.
//
// Replaces torch's DistributedDataParallel reducer over NCCL (reference run_multimodal_fcmf.py:169 init_process_group,
// :237-240 DDP wrap; run_pretraining_fcmf.py:196-199): one communicator per process (= per GPU), buckets of the flat
// gradient arena all-reduced IN PLACE on a caller-provided stream.
//
// RCCL is bound at RUN time (dlopen / dlsym): libfcmf_hip.so loads in processes without RCCL, and inside a PyTorch process
// it binds to the RCCL copy that process already holds (torch ships its own librccl.so: two different copies in one
// process would each keep their own topology / IPC state).  Only the type definitions of <rccl/rccl.h> are used at build time.
#include "common.h"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
  std::call_once(g_rccl_once, [] {
    // 1. a copy that is already loaded into this process (PyTorch's), 2. the ROCm installation's
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (g_rccl.handle) break;
    }
    if (!g_rccl.handle)
      for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
      }
    if (!g_rccl.handle) return;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.handle, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(g_rccl.handle, "ncclAllReduce"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce;
  });
  return g_rccl;
}

struct DpComm {
  ncclComm_t comm;
  int nranks, rank;
};
}  // namespace

extern "C" int fcmf_dp_unique_id(void* out) {
  if (!out) return FCMF_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == FCMF_DP_UNIQUE_ID_BYTES, "unique id size");
  const Rccl& r = rccl();
  if (!r.ok) return FCMF_ERR_COMM;
  return r.GetUniqueId(reinterpret_cast<ncclUniqueId*>(out)) == ncclSuccess ? FCMF_OK : FCMF_ERR_COMM;
}

extern "C" int fcmf_dp_comm_create(void** comm, const void* unique_id, int nranks, int rank) {
  if (!comm || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return FCMF_ERR_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return FCMF_ERR_COMM;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  ncclComm_t c = nullptr;
  if (r.CommInitRank(&c, nranks, id, rank) != ncclSuccess) return FCMF_ERR_COMM;   // (collective: every rank calls it)
  *comm = new DpComm{c, nranks, rank};
  return FCMF_OK;
}

extern "C" int fcmf_dp_comm_destroy(void* comm) {
  if (!comm) return FCMF_ERR_ARG;
  DpComm* d = reinterpret_cast<DpComm*>(comm);
  const ncclResult_t rc = rccl().CommDestroy(d->comm);
  delete d;
  return rc == ncclSuccess ? FCMF_OK : FCMF_ERR_COMM;
}

extern "C" int fcmf_dp_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, int average, void* stream) {
  if (!comm || !buf || count < 0) return FCMF_ERR_ARG;
  if (dtype != FCMF_F32 && dtype != FCMF_BF16) return FCMF_ERR_UNSUPPORTED;
  if (count == 0) return FCMF_OK;
  DpComm* d = reinterpret_cast<DpComm*>(comm);
  // ncclAvg: the mean is formed inside the collective's last reduction step -- no separate scaling pass over the bucket
  const ncclResult_t rc = rccl().AllReduce(buf, buf, (size_t)count, dtype == FCMF_F32 ? ncclFloat32 : ncclBfloat16,
                                           average ? ncclAvg : ncclSum, d->comm, reinterpret_cast<hipStream_t>(stream));
  return rc == ncclSuccess ? FCMF_OK : FCMF_ERR_COMM;
}
