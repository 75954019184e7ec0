// comm.hip -- gradient all-reduce in the C ABI: RCCL over xGMI.  Replaces the reference's parameter-server
// traffic (tf.train.replica_device_setter + async apply_gradients over gRPC, W/train.py:624-639,731-776) with the synchronous
// data-parallel mean SURVEY.md 8e defines.  One communicator per process (= per GPU); the caller distributes the 128-byte
// unique id of rank 0 (file, socket, MPI, torch store ...).
//
// RCCL is bound at run time (dlopen of the librccl the process already has, e.g. PyTorch's, else the ROCm one): the library
// itself carries no link-time dependency, and a host without RCCL gets YT8M_E_RCCL from these entry points only.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
  const char* names[] = {getenv("YT8M_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n) continue;
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
#define YT8M_SYM(field, sym) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, sym))
  YT8M_SYM(GetUniqueId, "ncclGetUniqueId");
  YT8M_SYM(CommInitRank, "ncclCommInitRank");
  YT8M_SYM(CommDestroy, "ncclCommDestroy");
  YT8M_SYM(AllReduce, "ncclAllReduce");
  YT8M_SYM(Broadcast, "ncclBroadcast");
  YT8M_SYM(ReduceScatter, "ncclReduceScatter");
  YT8M_SYM(AllGather, "ncclAllGather");
  YT8M_SYM(GetErrorString, "ncclGetErrorString");
  YT8M_SYM(CommCount, "ncclCommCount");
#undef YT8M_SYM
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast &&
              g_rccl.GetErrorString && g_rccl.CommCount;
}

int rccl_ready() {
  std::call_once(g_rccl_once, rccl_load);
  if (!g_rccl.ok) return yt8m::fail(YT8M_E_RCCL, "RCCL is not available (dlopen librccl.so failed: %s)", dlerror() ? "see YT8M_RCCL_LIB" : "");
  return YT8M_OK;
}

int rccl_fail(const char* what, ncclResult_t r) {
  snprintf(yt8m::g_err, sizeof(yt8m::g_err), "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return YT8M_E_RCCL;
}

struct Comm { ncclComm_t c; int rank, world; };

}  // namespace

using namespace yt8m;

extern "C" int yt8m_comm_unique_id(void* id_out) {
  YT8M_REQUIRE(id_out, YT8M_E_BADARG, "null id buffer (needs YT8M_COMM_ID_BYTES = 128 bytes)");
  int rc = rccl_ready();
  if (rc != YT8M_OK) return rc;
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  memcpy(id_out, &id, sizeof(id));
  return YT8M_OK;
}

extern "C" int yt8m_comm_init(int rank, int world, const void* unique_id, void** comm_out) {
  YT8M_REQUIRE(comm_out && unique_id, YT8M_E_BADARG, "null argument");
  YT8M_REQUIRE(world >= 1 && rank >= 0 && rank < world, YT8M_E_BADARG, "need 0 <= rank < world");
  int rc = rccl_ready();
  if (rc != YT8M_OK) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  Comm* c = new Comm{nullptr, rank, world};
  const ncclResult_t r = g_rccl.CommInitRank(&c->c, world, id, rank);   // binds the calling thread's current HIP device
  if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
  *comm_out = c;
  return YT8M_OK;
}

extern "C" int yt8m_comm_size(void* comm, int* rank, int* world) {
  YT8M_REQUIRE(comm, YT8M_E_BADARG, "null communicator");
  Comm* c = static_cast<Comm*>(comm);
  int n = 0;
  const ncclResult_t r = g_rccl.CommCount(c->c, &n);
  if (r != ncclSuccess) return rccl_fail("ncclCommCount", r);
  if (rank) *rank = c->rank;
  if (world) *world = n;
  return YT8M_OK;
}

// in place: buf <- mean over ranks (mean != 0) or sum (mean == 0) of the fp32 buffer; asynchronous on `stream`
extern "C" int yt8m_comm_allreduce_f32(void* comm, float* buf, int64_t n, int mean, yt8m_stream_t stream) {
  YT8M_REQUIRE(comm, YT8M_E_BADARG, "null communicator");
  YT8M_REQUIRE(n >= 0, YT8M_E_SHAPE, "negative count");
  if (n == 0) return YT8M_OK;
  YT8M_REQUIRE(buf, YT8M_E_BADARG, "null buffer");
  Comm* c = static_cast<Comm*>(comm);
  const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, mean ? ncclAvg : ncclSum, c->c, as_stream(stream));
  if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
  return YT8M_OK;
}

// The same result as yt8m_comm_allreduce_f32 as TWO collectives: reduce-scatter (every rank ends up owning the reduced values of
// its n / world slice) + all-gather of the slices, in place; the n % world tail goes through a small all-reduce.  Same bytes on
// the wire as a ring all-reduce, but the two halves are separate launches: a host can put work between them that only needs the
// rank's own slice (a sharded optimiser pass: clip + Adam on 1 / world of the parameters, then all-gather the PARAMETERS), and
// RCCL's algorithm choice for each half is independent of its all-reduce heuristics (on point-to-point xGMI a
// single ring is bound by one link).  `phase`: 0 = both, 1 = reduce-scatter (+ tail all-reduce) only, 2 = all-gather only.
extern "C" int yt8m_comm_allreduce_rsag_f32(void* comm, float* buf, int64_t n, int mean, int phase, yt8m_stream_t stream) {
  YT8M_REQUIRE(comm, YT8M_E_BADARG, "null communicator");
  YT8M_REQUIRE(n >= 0 && phase >= 0 && phase <= 2, YT8M_E_SHAPE, "negative count / bad phase");
  if (n == 0) return YT8M_OK;
  YT8M_REQUIRE(buf, YT8M_E_BADARG, "null buffer");
  YT8M_REQUIRE(g_rccl.ReduceScatter && g_rccl.AllGather, YT8M_E_RCCL, "this RCCL has no ncclReduceScatter / ncclAllGather");
  Comm* c = static_cast<Comm*>(comm);
  const int64_t per = n / c->world, tail = n - per * c->world;
  const ncclRedOp_t op = mean ? ncclAvg : ncclSum;
  hipStream_t s = as_stream(stream);
  if (phase != 2) {
    if (per > 0) {
      const ncclResult_t r = g_rccl.ReduceScatter(buf, buf + (int64_t)c->rank * per, (size_t)per, ncclFloat32, op, c->c, s);
      if (r != ncclSuccess) return rccl_fail("ncclReduceScatter", r);
    }
    if (tail > 0) {
      const ncclResult_t r = g_rccl.AllReduce(buf + per * c->world, buf + per * c->world, (size_t)tail, ncclFloat32, op, c->c, s);
      if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    }
  }
  if (phase != 1 && per > 0) {
    const ncclResult_t r = g_rccl.AllGather(buf + (int64_t)c->rank * per, buf, (size_t)per, ncclFloat32, c->c, s);
    if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
  }
  return YT8M_OK;
}

extern "C" int yt8m_comm_allreduce_mean(void* comm, float* buf, int64_t n, yt8m_stream_t stream) {
  return yt8m_comm_allreduce_f32(comm, buf, n, 1, stream);
}

// every rank starts from rank `root`'s parameters (W/train.py: chief initialises, workers wait)
extern "C" int yt8m_comm_broadcast_f32(void* comm, float* buf, int64_t n, int root, yt8m_stream_t stream) {
  YT8M_REQUIRE(comm, YT8M_E_BADARG, "null communicator");
  YT8M_REQUIRE(n >= 0, YT8M_E_SHAPE, "negative count");
  if (n == 0) return YT8M_OK;
  YT8M_REQUIRE(buf, YT8M_E_BADARG, "null buffer");
  Comm* c = static_cast<Comm*>(comm);
  YT8M_REQUIRE(root >= 0 && root < c->world, YT8M_E_BADARG, "root out of range");
  const ncclResult_t r = g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->c, as_stream(stream));
  if (r != ncclSuccess) return rccl_fail("ncclBroadcast", r);
  return YT8M_OK;
}

extern "C" int yt8m_comm_destroy(void* comm) {
  if (!comm) return YT8M_OK;
  Comm* c = static_cast<Comm*>(comm);
  const ncclResult_t r = g_rccl.CommDestroy(c->c);
  delete c;
  if (r != ncclSuccess) return rccl_fail("ncclCommDestroy", r);
  return YT8M_OK;
}
