// Gradient exchange of a data-parallel run (the reference trains on one GPU, srgan_train.py:58-61): one
// process per GPU, a native RCCL communicator over xGMI.  librccl is opened at run time (dlopen), so libdbm.so loads on
// hosts without it and a single-GPU run never touches it.  The collectives are enqueued on library stream chain[1]
// (idle while the backward passes run), bucket by bucket as the weight-gradient launches of a layer group finish:
// no fifth busy stream, no host synchronisation.  A caller hook (dbm_comm_set_hook) stands in for RCCL where RCCL
// cannot run (several ranks sharing one GPU in a test).
#include "model.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  if (r.h) return r;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.h) break;
  }
  if (!r.h) throw DbmError(6, std::string("librccl could not be opened: ") + dlerror());
  auto sym = [&](const char* n) {
    void* p = dlsym(r.h, n);
    if (!p) throw DbmError(6, std::string("librccl lacks ") + n);
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  return r;
}

#define DBM_NCCL(expr)                                                                                        \
  do {                                                                                                        \
    ncclResult_t _r = (expr);                                                                                 \
    if (_r != ncclSuccess)                                                                                    \
      throw DbmError(6, std::string(#expr) + ": " + rccl().GetErrorString(_r) + " @" + __FILE__ + ":" +       \
                            std::to_string(__LINE__));                                                        \
  } while (0)

}  // namespace

void dbm_comm_unique_id_impl(void* out128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  DBM_NCCL(rccl().GetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
}

void dbm_ctx::comm_init(int rank, int world, const void* id128) {
  DBM_CHECK(world >= 1 && rank >= 0 && rank < world, "dbm_comm_init: bad rank / world");
  comm_destroy();
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  DBM_HIP(hipSetDevice(device));
  ncclComm_t c = nullptr;
  DBM_NCCL(rccl().CommInitRank(&c, world, id, rank));
  nccl_comm = c;
  comm_rank = rank;
  comm_world = world;
  comm_hook = nullptr;
}

void dbm_ctx::comm_set_hook(int rank, int world, void (*fn)(void*, float*, size_t, void*), void* user) {
  DBM_CHECK(world >= 1 && rank >= 0 && rank < world, "dbm_comm_set_hook: bad rank / world");
  DBM_CHECK(world == 1 || fn != nullptr, "dbm_comm_set_hook: a hook is required for world > 1");
  comm_destroy();
  comm_rank = rank;
  comm_world = world;
  comm_hook = world > 1 ? fn : nullptr;
  comm_user = user;
}

void dbm_ctx::comm_destroy() {
  if (nccl_comm) {
    (void)hipStreamSynchronize(chain[1]);
    (void)hipStreamSynchronize(stream);
    (void)rccl().CommDestroy((ncclComm_t)nccl_comm);
    nccl_comm = nullptr;
  }
  comm_hook = nullptr;
  comm_world = 1;
  comm_rank = 0;
}

// in-place sum over ranks of the float ranges {p[i], n[i]} on stream `on` (one fused RCCL group)
void dbm_ctx::comm_allreduce(float* const* p, const size_t* n, int nranges, hipStream_t on) {
  if (!comm_active()) return;
  if (nccl_comm) {
    if (nranges > 1) DBM_NCCL(rccl().GroupStart());
    for (int i = 0; i < nranges; ++i)
      if (n[i]) DBM_NCCL(rccl().AllReduce(p[i], p[i], n[i], ncclFloat32, ncclSum, (ncclComm_t)nccl_comm, on));
    if (nranges > 1) DBM_NCCL(rccl().GroupEnd());
  } else {
    for (int i = 0; i < nranges; ++i)
      if (n[i]) comm_hook(comm_user, p[i], n[i], (void*)on);
  }
  comm_bytes += [&] { size_t t = 0; for (int i = 0; i < nranges; ++i) t += n[i] * sizeof(float); return t; }();
  comm_calls += 1;
}

// sync_batch_stats collective (BatchNorm / RaGAN sums) on the main stream: the caller's hook, else the native communicator
void dbm_ctx::allreduce(float* dev, int n) {
  if (sync_fn) {
    sync_fn(sync_user, dev, n);
    return;
  }
  float* p = dev;
  size_t m = (size_t)n;
  comm_allreduce(&p, &m, 1, stream);
}

void dbm_ctx::comm_broadcast(float* p, size_t n, int root, hipStream_t on) {
  if (comm_world <= 1) return;
  DBM_CHECK(nccl_comm != nullptr, "dbm_comm_broadcast needs the native communicator (dbm_comm_init)");
  DBM_NCCL(rccl().Broadcast(p, p, n, ncclFloat32, root, (ncclComm_t)nccl_comm, on));
}

// Bucket of the gradient arena whose producers have all been enqueued on `producer`: the exchange stream (chain[1])
// waits for them and takes the all-reduce; nobody else waits (comm_join at the end of the step).
void dbm_ctx::comm_bucket(float* const* p, const size_t* n, int nranges, hipStream_t producer) {
  if (!comm_active()) return;
  hipStream_t cs = chain[1];
  if (producer != cs) {
    if (!ev_comm) DBM_HIP(hipEventCreateWithFlags(&ev_comm, hipEventDisableTiming));
    DBM_HIP(hipEventRecord(ev_comm, producer));
    DBM_HIP(hipStreamWaitEvent(cs, ev_comm, 0));
  }
  DBM_MARK(cs, "  comm:bucket_begin");
  comm_allreduce(p, n, nranges, cs);
  DBM_MARK(cs, "  comm:bucket_end");
}

void dbm_ctx::comm_join(hipStream_t consumer) {
  if (!comm_active()) return;
  if (!ev_comm_done) DBM_HIP(hipEventCreateWithFlags(&ev_comm_done, hipEventDisableTiming));
  DBM_HIP(hipEventRecord(ev_comm_done, chain[1]));
  DBM_HIP(hipStreamWaitEvent(consumer, ev_comm_done, 0));
}
