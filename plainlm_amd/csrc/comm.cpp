// RCCL gradient exchange behind the C ABI (replaces DDP's reducer: engine/engine.py:64-65,104-105).
// One communicator per process, one process per GPU; collectives run on the caller's side stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include "../../include/plainlm_hip.h"

void plm_set_error(const char* fmt, ...);

struct plm_comm {
  ncclComm_t comm;
  int rank, world, device;
};

static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");

#define PLM_NCCL(call, what)                                                  \
  do {                                                                        \
    ncclResult_t r__ = (call);                                                \
    if (r__ != ncclSuccess) {                                                 \
      plm_set_error("%s: %s", what, ncclGetErrorString(r__));                 \
      return PLM_E_COMM;                                                      \
    }                                                                         \
  } while (0)

extern "C" int plm_comm_unique_id(uint8_t uid[128]) {
  if (!uid) {
    plm_set_error("plm_comm_unique_id: null pointer");
    return PLM_E_INVALID;
  }
  ncclUniqueId id;
  PLM_NCCL(ncclGetUniqueId(&id), "ncclGetUniqueId");
  memcpy(uid, &id, 128);
  return PLM_OK;
}

static void capped_config(ncclConfig_t* cfg, int max_ctas) {
  if (max_ctas > 0) {
    cfg->minCTAs = 1;
    cfg->maxCTAs = max_ctas;
  }
}

extern "C" int plm_comm_init(plm_comm_t** out, const uint8_t uid[128], int rank, int world_size, int device) {
  return plm_comm_init_capped(out, uid, rank, world_size, device, 0);
}

extern "C" int plm_comm_split(plm_comm_t* parent, plm_comm_t** child, int max_ctas) {
  if (!parent || !child) {
    plm_set_error("plm_comm_split: null pointer");
    return PLM_E_INVALID;
  }
  hipError_t e = hipSetDevice(parent->device);
  if (e != hipSuccess) {
    plm_set_error("plm_comm_split: hipSetDevice(%d): %s", parent->device, hipGetErrorString(e));
    return PLM_E_HIP;
  }
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  capped_config(&cfg, max_ctas);
  plm_comm* c = new plm_comm{nullptr, parent->rank, parent->world, parent->device};
  ncclResult_t r = ncclCommSplit(parent->comm, /*color*/ 0, /*key*/ parent->rank, &c->comm, &cfg);
  if (r != ncclSuccess || c->comm == nullptr) {
    plm_set_error("ncclCommSplit: %s", ncclGetErrorString(r));
    delete c;
    return PLM_E_COMM;
  }
  *child = c;
  return PLM_OK;
}

extern "C" int plm_comm_init_capped(plm_comm_t** out, const uint8_t uid[128], int rank, int world_size, int device, int max_ctas) {
  if (!out || !uid || world_size < 1 || rank < 0 || rank >= world_size) {
    plm_set_error("plm_comm_init: bad arguments (rank=%d world=%d)", rank, world_size);
    return PLM_E_INVALID;
  }
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    plm_set_error("plm_comm_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return PLM_E_HIP;
  }
  ncclUniqueId id;
  memcpy(&id, uid, 128);
  plm_comm* c = new plm_comm{nullptr, rank, world_size, device};
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  capped_config(&cfg, max_ctas);
  ncclResult_t r = max_ctas > 0 ? ncclCommInitRankConfig(&c->comm, world_size, id, rank, &cfg) : ncclCommInitRank(&c->comm, world_size, id, rank);
  if (r != ncclSuccess) {
    plm_set_error("ncclCommInitRank%s: %s", max_ctas > 0 ? "Config" : "", ncclGetErrorString(r));
    delete c;
    return PLM_E_COMM;
  }
  *out = c;
  return PLM_OK;
}

extern "C" int plm_comm_destroy(plm_comm_t* c) {
  if (!c) return PLM_OK;
  ncclResult_t r = ncclCommDestroy(c->comm);
  delete c;
  if (r != ncclSuccess) {
    plm_set_error("ncclCommDestroy: %s", ncclGetErrorString(r));
    return PLM_E_COMM;
  }
  return PLM_OK;
}

extern "C" int plm_comm_allreduce_avg_f32(plm_comm_t* c, float* buf, int64_t count, void* stream) {
  if (!c || !buf || count < 0) {
    plm_set_error("plm_comm_allreduce_avg_f32: bad arguments");
    return PLM_E_INVALID;
  }
  if (count == 0) return PLM_OK;
  PLM_NCCL(ncclAllReduce(buf, buf, (size_t)count, ncclFloat32, ncclAvg, c->comm, (hipStream_t)stream), "ncclAllReduce");
  return PLM_OK;
}

// The same mean as plm_comm_allreduce_avg_f32, spelled as reduce-scatter + all-gather in place (on a fully connected
// xGMI node a one-hop reduce-scatter / all-gather pair uses all 7 links of a GPU at once, a single ring one link each way; which of the
// two RCCL's own ncclAllReduce picks depends on its tuning tables).  The span is cut into `world` equal chunks (rank r reduces chunk r);
// the count % world elements left over go through a small all-reduce.  Opt-in (PLM_COMM_ALGO=rsag): no multi-GPU run has compared them.
extern "C" int plm_comm_rsag_avg_f32(plm_comm_t* c, float* buf, int64_t count, void* stream) {
  if (!c || !buf || count < 0) {
    plm_set_error("plm_comm_rsag_avg_f32: bad arguments");
    return PLM_E_INVALID;
  }
  if (count == 0) return PLM_OK;
  const int64_t chunk = count / c->world, tail = count - chunk * c->world;
  hipStream_t s = (hipStream_t)stream;
  if (chunk > 0) {
    float* mine = buf + (int64_t)c->rank * chunk;  // in-place forms: recvbuff = sendbuff + rank * recvcount / sendbuff = recvbuff + rank * sendcount
    PLM_NCCL(ncclReduceScatter(buf, mine, (size_t)chunk, ncclFloat32, ncclAvg, c->comm, s), "ncclReduceScatter");
    PLM_NCCL(ncclAllGather(mine, buf, (size_t)chunk, ncclFloat32, c->comm, s), "ncclAllGather");
  }
  if (tail > 0) PLM_NCCL(ncclAllReduce(buf + chunk * c->world, buf + chunk * c->world, (size_t)tail, ncclFloat32, ncclAvg, c->comm, s), "ncclAllReduce (tail)");
  return PLM_OK;
}

extern "C" int plm_comm_broadcast_f32(plm_comm_t* c, float* buf, int64_t count, int root, void* stream) {
  if (!c || !buf || count < 0 || root < 0 || root >= c->world) {
    plm_set_error("plm_comm_broadcast_f32: bad arguments");
    return PLM_E_INVALID;
  }
  if (count == 0) return PLM_OK;
  PLM_NCCL(ncclBroadcast(buf, buf, (size_t)count, ncclFloat32, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
  return PLM_OK;
}
