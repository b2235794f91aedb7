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
    (void)hipStreamSynchronize(chain[0]);
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

// Bucket of the gradient arena whose producers have all been enqueued on `producer`: the exchange stream waits for them
// and takes the all-reduce; nobody else waits (comm_join at the end of the step).  The exchange stream is chain[1] in the
// two step calls (idle while the backward passes run) and chain[0] in dbm_train_iteration, where chain[1] carries the
// generator's own forward and backward pass (dbm_ctx::comm_stream).  With comm_defer set the collective is NOT enqueued
// yet -- the producers' event is recorded now, the all-reduce goes out with comm_flush(): a bucket that becomes ready
// while its exchange stream still receives the kernels of a backward pass must not cut that pass in two.
void dbm_ctx::comm_bucket(float* const* p, const size_t* n, int nranges, hipStream_t producer) {
  if (!comm_active()) return;
  PendingBucket b;
  b.nranges = nranges < 2 ? nranges : 2;
  for (int i = 0; i < b.nranges; ++i) { b.p[i] = p[i]; b.n[i] = n[i]; }
  b.ev = nullptr;
  hipStream_t cs = comm_stream ? comm_stream : chain[1];
  if (producer != cs || comm_defer) {
    if (comm_ev_used == comm_ev_pool.size()) {
      hipEvent_t e;
      DBM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      comm_ev_pool.push_back(e);
    }
    b.ev = comm_ev_pool[comm_ev_used++];
    DBM_HIP(hipEventRecord(b.ev, producer));
  }
  comm_pending.push_back(b);
  if (!comm_defer) comm_flush();
}

void dbm_ctx::comm_flush() {
  hipStream_t cs = comm_stream ? comm_stream : chain[1];
  for (const PendingBucket& b : comm_pending) {
    if (b.ev) DBM_HIP(hipStreamWaitEvent(cs, b.ev, 0));
    DBM_MARK(cs, "  comm:bucket_begin");
    comm_allreduce(b.p, b.n, b.nranges, cs);
    DBM_MARK(cs, "  comm:bucket_end");
  }
  comm_pending.clear();
}

void dbm_ctx::comm_join(hipStream_t consumer) {
  if (!comm_active()) return;
  comm_flush();
  comm_ev_used = 0;  // (every event of the pool has been waited for by the exchange stream: free for the next pass)
  hipStream_t cs = comm_stream ? comm_stream : chain[1];
  if (!ev_comm_done) DBM_HIP(hipEventCreateWithFlags(&ev_comm_done, hipEventDisableTiming));
  DBM_HIP(hipEventRecord(ev_comm_done, cs));
  DBM_HIP(hipStreamWaitEvent(consumer, ev_comm_done, 0));
}
