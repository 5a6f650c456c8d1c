// rbd_comm.hip — the one exchange step of the path when the batch is sharded over the GPUs of a node (SURVEY.md §8 e, BASELINE
// configs[3]): an RCCL all-gather (or gather to one rank) of each rank's shard of DynamicsResult.v̇ over xGMI, behind the C ABI so that
// a non-Python caller (the Julia shim) can run it without torch.distributed.  States are independent: there is no collective inside
// the dynamics itself.  One process per GPU; the caller distributes the 128-byte unique id among its ranks (as with ncclGetUniqueId).
// librccl is opened on first use (dlopen): single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <mutex>
#include "rbd_hip.h"

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
void rccl_open(Rccl& R);
Rccl& rccl() {
  static Rccl R;
  static std::once_flag once;  // two threads creating communicators at once must not race on the function pointers
  std::call_once(once, [] { rccl_open(R); });
  return R;
}
void rccl_open(Rccl& R) {
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (R.h) break;
  }
  if (!R.h) return;
  auto sym = [&](const char* n) { return dlsym(R.h, n); };
  R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
  R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
  R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
  R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
  R.Send = (decltype(R.Send))sym("ncclSend");
  R.Recv = (decltype(R.Recv))sym("ncclRecv");
  R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
  R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
  R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllGather && R.Send && R.Recv && R.GroupStart && R.GroupEnd;
}
thread_local std::string g_comm_error;
}  // namespace

struct rbd_comm {
  ncclComm_t comm = nullptr;
  int32_t world = 0, rank = 0, device = 0;
  int64_t* d_counts = nullptr;  // RBD_COMM_CHECK=1: world + world * world counts (this rank's, then everybody's), allocated with the communicator
};

extern "C" {

const char* rbd_comm_last_error(void) { return g_comm_error.c_str(); }

int rbd_comm_unique_id(void* id128) {
  if (!id128) return RBD_ERR_INVALID_ARGUMENT;
  Rccl& R = rccl();
  if (!R.ok) { g_comm_error = "librccl.so could not be opened"; return RBD_ERR_UNSUPPORTED; }
  ncclUniqueId id;
  const ncclResult_t r = R.GetUniqueId(&id);
  if (r != ncclSuccess) { g_comm_error = R.GetErrorString ? R.GetErrorString(r) : "ncclGetUniqueId failed"; return RBD_ERR_HIP; }
  static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
  memcpy(id128, &id, sizeof id);
  return RBD_OK;
}

int rbd_comm_create(const void* id128, int32_t world, int32_t rank, int32_t device, rbd_comm_t** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return RBD_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  Rccl& R = rccl();
  if (!R.ok) { g_comm_error = "librccl.so could not be opened"; return RBD_ERR_UNSUPPORTED; }
  if (hipSetDevice(device) != hipSuccess) return RBD_ERR_NO_DEVICE;
  rbd_comm* c = new (std::nothrow) rbd_comm();
  if (!c) return RBD_ERR_OUT_OF_MEMORY;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  const ncclResult_t r = R.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { g_comm_error = R.GetErrorString ? R.GetErrorString(r) : "ncclCommInitRank failed"; delete c; return RBD_ERR_HIP; }
  c->world = world; c->rank = rank; c->device = device;
  if (hipMalloc((void**)&c->d_counts, sizeof(int64_t) * ((size_t)world + (size_t)world * world)) != hipSuccess) { (void)hipGetLastError(); c->d_counts = nullptr; }
  *out = c;
  return RBD_OK;
}

int rbd_comm_destroy(rbd_comm_t* c) {
  if (!c) return RBD_OK;
  if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
  if (c->d_counts) (void)hipFree(c->d_counts);
  delete c;
  return RBD_OK;
}

int rbd_comm_info(const rbd_comm_t* c, int32_t* world, int32_t* rank) {
  if (!c) return RBD_ERR_INVALID_ARGUMENT;
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  return RBD_OK;
}

// gathered[r * count .. (r+1) * count) = rank r's shard (count scalars of dtype); root < 0: on every rank (ncclAllGather);
// root >= 0: on that rank only (grouped ncclSend / ncclRecv).  Asynchronous on `stream` (a hipStream_t; NULL = default stream).
int rbd_gather(rbd_comm_t* c, int32_t dtype, const void* shard, void* gathered, int64_t count, int32_t root, void* stream) {
  if (!c || !shard || count < 0 || (dtype != RBD_F64 && dtype != RBD_F32) || root >= c->world) return RBD_ERR_INVALID_ARGUMENT;
  if ((root < 0 || root == c->rank) && !gathered) return RBD_ERR_INVALID_ARGUMENT;
  Rccl& R = rccl();
  if (!R.ok) { g_comm_error = "librccl.so could not be opened"; return RBD_ERR_UNSUPPORTED; }
  const ncclDataType_t t = dtype == RBD_F64 ? ncclDouble : ncclFloat;
  const size_t es = dtype == RBD_F64 ? 8 : 4;
  hipStream_t s = (hipStream_t)stream;
  if (hipSetDevice(c->device) != hipSuccess) return RBD_ERR_NO_DEVICE;
  ncclResult_t r = ncclSuccess;
  if (root < 0) {
    r = R.AllGather(shard, gathered, (size_t)count, t, c->comm, s);
  } else {
    r = R.GroupStart();
    if (r == ncclSuccess) r = R.Send(shard, (size_t)count, t, root, c->comm, s);
    if (c->rank == root)
      for (int p = 0; p < c->world && r == ncclSuccess; ++p) r = R.Recv((char*)gathered + (size_t)p * count * es, (size_t)count, t, p, c->comm, s);
    const ncclResult_t e = R.GroupEnd();
    if (r == ncclSuccess) r = e;
  }
  if (r != ncclSuccess) { g_comm_error = R.GetErrorString ? R.GetErrorString(r) : "rccl call failed"; return RBD_ERR_HIP; }
  return RBD_OK;
}

// ... with shards of DIFFERENT sizes (a batch that does not divide by the number of ranks: shard_range gives the first B % world ranks one state more):
// counts[r] = rank r's scalars, gathered = the shards back to back in rank order.  Grouped ncclSend / ncclRecv (there is no ragged all-gather in RCCL): to the
// root only, or — root < 0 — from every rank to every rank.
int rbd_gatherv(rbd_comm_t* c, int32_t dtype, const void* shard, void* gathered, const int64_t* counts, int32_t root, void* stream) {
  if (!c || !counts || (dtype != RBD_F64 && dtype != RBD_F32) || root >= c->world) return RBD_ERR_INVALID_ARGUMENT;
  for (int p = 0; p < c->world; ++p)
    if (counts[p] < 0) return RBD_ERR_INVALID_ARGUMENT;
  if (counts[c->rank] > 0 && !shard) return RBD_ERR_INVALID_ARGUMENT;
  if ((root < 0 || root == c->rank) && !gathered) return RBD_ERR_INVALID_ARGUMENT;
  Rccl& R = rccl();
  if (!R.ok) { g_comm_error = "librccl.so could not be opened"; return RBD_ERR_UNSUPPORTED; }
  const ncclDataType_t t = dtype == RBD_F64 ? ncclDouble : ncclFloat;
  const size_t es = dtype == RBD_F64 ? 8 : 4;
  hipStream_t s = (hipStream_t)stream;
  if (hipSetDevice(c->device) != hipSuccess) return RBD_ERR_NO_DEVICE;
  // RBD_COMM_CHECK=1 (set on EVERY rank or on none): the ranks compare their counts first — one all-gather of world values each, a wait for the stream — and a rank
  // whose neighbours disagree with it returns RBD_ERR_DIMENSION_MISMATCH instead of waiting for a send that never comes (round-5 advice: disagreeing counts hang
  // the group like mismatched ncclSend / ncclRecv sizes).  Off by default: the gather of v̇ is asynchronous on the caller's stream.
  if (const char* e = getenv("RBD_COMM_CHECK")) {
    if (e[0] == '1' && c->d_counts) {
      const size_t w = (size_t)c->world;
      std::vector<int64_t> all(w * w);
      if (hipMemcpyAsync(c->d_counts, counts, sizeof(int64_t) * w, hipMemcpyHostToDevice, s) != hipSuccess) return RBD_ERR_HIP;
      const ncclResult_t rc = R.AllGather(c->d_counts, c->d_counts + w, w, ncclInt64, c->comm, s);
      if (rc != ncclSuccess) { g_comm_error = R.GetErrorString ? R.GetErrorString(rc) : "rccl call failed"; return RBD_ERR_HIP; }
      if (hipMemcpyAsync(all.data(), c->d_counts + w, sizeof(int64_t) * w * w, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return RBD_ERR_HIP;
      for (size_t p = 0; p < w; ++p)
        if (memcmp(&all[p * w], counts, sizeof(int64_t) * w) != 0) {
          g_comm_error = "rbd_gatherv: rank " + std::to_string(p) + " passed other counts than rank " + std::to_string(c->rank);
          return RBD_ERR_DIMENSION_MISMATCH;
        }
    }
  }
  ncclResult_t r = R.GroupStart();
  const int64_t mine = counts[c->rank];
  for (int dst = 0; dst < c->world && r == ncclSuccess; ++dst)  // (an empty shard is neither sent nor waited for: both sides read the same counts)
    if (mine > 0 && (root < 0 || dst == root)) r = R.Send(shard, (size_t)mine, t, dst, c->comm, s);
  if (root < 0 || root == c->rank) {
    size_t off = 0;
    for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
      if (counts[p] > 0) r = R.Recv((char*)gathered + off * es, (size_t)counts[p], t, p, c->comm, s);
      off += (size_t)counts[p];
    }
  }
  const ncclResult_t e = R.GroupEnd();
  if (r == ncclSuccess) r = e;
  if (r != ncclSuccess) { g_comm_error = R.GetErrorString ? R.GetErrorString(r) : "rccl call failed"; return RBD_ERR_HIP; }
  return RBD_OK;
}

}  // extern "C"
