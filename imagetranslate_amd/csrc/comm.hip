// RCCL behind the C ABI: the gradient exchange of the data-parallel train step (torch DistributedDataParallel over NCCL at
// src/train_image_mt.py:72-76, process-group bootstrap at src/utils.py:93-97) as three plain entry points -- init /
// all-reduce / destroy -- plus the broadcast DDP's constructor does and the unique-id hand-shake RCCL needs.
// One process per GPU, one communicator per process; ranks exchange the 128-byte unique id out of band (the Python side
// uses the torch.distributed store it already has; a C caller any side channel).  librccl is opened at first use
// (dlopen), so single-GPU use of libimt_hip.so has no load-time dependency on it.  Calls are asynchronous in `stream`
// like every other entry point; in-place sum over the ranks (the 1/world_size lives in imt_clip_adam's grad_scale).
#include <dlfcn.h>
#include "common.hpp"

namespace {

// the handful of RCCL declarations used (ABI of rccl.h: ncclResult_t = int, 0 = success; ncclDataType_t / ncclRedOp_t enums)
struct UniqueId { char internal[128]; };
typedef void* Comm;
enum { NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_SUM = 0 };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  BroadcastFn broadcast = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  bool tried = false;
  bool load() {
    if (tried) return handle != nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) return false;
    get_unique_id = (GetUniqueIdFn)dlsym(handle, "ncclGetUniqueId");
    comm_init_rank = (CommInitRankFn)dlsym(handle, "ncclCommInitRank");
    all_reduce = (AllReduceFn)dlsym(handle, "ncclAllReduce");
    broadcast = (BroadcastFn)dlsym(handle, "ncclBroadcast");
    comm_destroy = (CommDestroyFn)dlsym(handle, "ncclCommDestroy");
    error_string = (GetErrorStringFn)dlsym(handle, "ncclGetErrorString");
    if (!get_unique_id || !comm_init_rank || !all_reduce || !broadcast || !comm_destroy) { dlclose(handle); handle = nullptr; }
    return handle != nullptr;
  }
};
Rccl g_rccl;

int rccl_check(int rc, const char* what) {
  if (rc == 0) return IMT_OK;
  imt_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
  return IMT_ERR_LAUNCH;
}
int nccl_type(int dtype) { return dtype == IMT_BF16 ? NCCL_BFLOAT16 : NCCL_FLOAT32; }

}  // namespace

#define IMT_NEED_RCCL(what) IMT_CHECK_ARG(g_rccl.load(), what ": librccl could not be opened (is ROCm's RCCL installed?)")

extern "C" int imt_comm_unique_id_bytes(void) { return (int)sizeof(UniqueId); }

extern "C" int imt_comm_get_unique_id(void* host_id_out) {
  IMT_CHECK_ARG(host_id_out, "comm_get_unique_id: null pointer");
  IMT_NEED_RCCL("comm_get_unique_id");
  return rccl_check(g_rccl.get_unique_id(reinterpret_cast<UniqueId*>(host_id_out)), "ncclGetUniqueId");
}

extern "C" int imt_comm_init(const void* host_unique_id, int world_size, int rank, void** comm_out) {
  IMT_CHECK_ARG(host_unique_id && comm_out, "comm_init: null pointer");
  IMT_CHECK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "comm_init: rank %d outside [0, %d)", rank, world_size);
  IMT_NEED_RCCL("comm_init");
  UniqueId id;
  memcpy(&id, host_unique_id, sizeof(id));
  Comm c = nullptr;
  const int rc = rccl_check(g_rccl.comm_init_rank(&c, world_size, id, rank), "ncclCommInitRank");
  if (rc != IMT_OK) return rc;
  *comm_out = c;
  return IMT_OK;
}

extern "C" int imt_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, void* stream) {
  IMT_CHECK_ARG(comm && (buf || count == 0) && count >= 0, "comm_allreduce: bad arguments");
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "comm_allreduce: bad dtype");
  if (count == 0) return IMT_OK;
  IMT_NEED_RCCL("comm_allreduce");
  ImtProfScope prof("rccl_allreduce", 0.0, (double)count * (dtype == IMT_BF16 ? 2 : 4), (hipStream_t)stream);
  return rccl_check(g_rccl.all_reduce(buf, buf, (size_t)count, nccl_type(dtype), NCCL_SUM, (Comm)comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int imt_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream) {
  IMT_CHECK_ARG(comm && (buf || count == 0) && count >= 0 && root >= 0, "comm_broadcast: bad arguments");
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "comm_broadcast: bad dtype");
  if (count == 0) return IMT_OK;
  IMT_NEED_RCCL("comm_broadcast");
  return rccl_check(g_rccl.broadcast(buf, buf, (size_t)count, nccl_type(dtype), root, (Comm)comm, (hipStream_t)stream), "ncclBroadcast");
}

extern "C" int imt_comm_destroy(void* comm) {
  if (!comm) return IMT_OK;
  IMT_NEED_RCCL("comm_destroy");
  return rccl_check(g_rccl.comm_destroy((Comm)comm), "ncclCommDestroy");
}
