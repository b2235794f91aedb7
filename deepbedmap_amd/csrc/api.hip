// extern "C" surface of libdbm.so (include/dbm.h).  No exception crosses the boundary.
#include "model.h"
#include <cmath>

static thread_local std::string g_last_error;

static void run_deferred_eval(dbm_ctx* c, hipStream_t on);   // (dbm_ctx::DeferredEval; defined next to dbm_train_iteration)
#define DBM_API_BEGIN_NOFLUSH(ctxptr) \
  dbm_ctx* _ectx = (ctxptr);  \
  (void)_ectx;                \
  try {
// every entry point but dbm_train_iteration first enqueues a pending deferred eval-mode pass on the main stream
#define DBM_API_BEGIN(ctxptr) \
  DBM_API_BEGIN_NOFLUSH(ctxptr) \
  if (_ectx && _ectx->deferred.pending) run_deferred_eval(_ectx, _ectx->stream);
// ---- a persistent trunk kernel that gives up (bounded spins: another process starving the GPU, a partitioned device) ----
// It raises the context's error word (host-mapped) and a STICKY device flag.  While the flag is up, every kernel that commits
// training state is a no-op: the optimizer launches (gated ONCE per launch by adam_gate_kernel, so an update is all or
// nothing) and BatchNorm's running-average writes -- no parameter, moment or running statistic ever absorbs an invalid
// pass.  The condition is HANDLED only at the entry of the step entry points (dbm_train_iteration, dbm_discriminator_step,
// dbm_generator_step, dbm_adam_update) and by dbm_check_timeout: nothing of the observing call has been enqueued yet, the
// device is drained, the optimizers' step counters take back the launches that were no-ops (reported: dbm_timeout_info), the
// persistent kernels are switched off for DBM_TRUNK_REARM iterations (default 64, doubled by every further event), and the
// call returns status 7 WITHOUT having done anything -- the caller re-issues it.  What is lost are the updates of the
// minibatches that were queued between the event and its observation (their count is reported; their metric rows are
// invalid).  Other entry points (copies, tensor reads) neither observe nor clear the condition.  In a data-parallel run the
// ranks can no longer be kept identical by a local retry: status 8, fatal.
static int trunk_rearm_after() {
  static const int v = getenv("DBM_TRUNK_REARM") ? atoi(getenv("DBM_TRUNK_REARM")) : 64;
  return v;
}
static void dbm_handle_persistent_timeout(dbm_ctx* c) {
  (void)hipDeviceSynchronize();
  c->timeout_events += 1;
  c->timeout_skipped[0] = c->timeout_skipped[1] = 0;
  for (dbm_model* m : c->models) {  // optimizer launches that found the condition up did nothing: take their step counts back
    int n = 0;
    if (m->type == 0) static_cast<Generator*>(m)->graph_version = -1;  // retained / prefetched passes are void
    // ... and so is whatever a backward pass has summed into the gradient arenas since the event (the observing call may be a
    // host-synchronising forward, long before the update that would apply them): marked until the next cleargrads
    if (m->grads_touched) m->grads_void = true;
    if (m->is_view || !m->d_adam_skipped) continue;
    if (hipMemcpy(&n, m->d_adam_skipped, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && n > 0) {
      m->adam_t -= n;
      c->timeout_skipped[m->type == 1 ? 0 : 1] += n;
      (void)hipMemset(m->d_adam_skipped, 0, sizeof(int));
    }
  }
  *(volatile int*)c->dev_err = 0;
  if (c->dev_err_flag) (void)hipMemset(c->dev_err_flag, 0, sizeof(int));
  (void)hipDeviceSynchronize();
  const long pause = trunk_rearm_after() <= 0 ? -1 : (long)trunk_rearm_after() << (c->timeout_events > 16 ? 16 : c->timeout_events - 1);
  g_trunk_fused_off = true;
  g_trunk_local_off = true;  // (whatever the cause was: the re-armed kernels exchange through agent-scope stores only)
  g_trunk_rearm_at = pause < 0 ? -1 : g_step_serial + pause;
  fprintf(stderr, "libdbm: a persistent trunk kernel gave up waiting for a neighbouring workgroup (event %ld); %d discriminator / %d "
                  "generator optimizer updates were skipped; the layer-by-layer trunk path is used for the next %ld iterations\n",
          c->timeout_events, c->timeout_skipped[0], c->timeout_skipped[1], pause);
}

// the gradient arena of `m` (shared with its views) has just been cleared on the stream: whatever a void pass left there is gone
static void mark_grads_cleared(dbm_model* m) {
  for (dbm_model* o : m->ctx->models)
    if (o->grads == m->grads) { o->grads_void = false; o->grads_touched = false; }
}

// entry of a step entry point: re-arm the persistent kernels when their pause is over, then observe the condition
static void dbm_step_entry(dbm_ctx* c, bool counts_as_iteration) {
  if (counts_as_iteration) g_step_serial += 1;
  if (g_trunk_fused_off && g_trunk_rearm_at >= 0 && g_step_serial >= g_trunk_rearm_at) {
    g_trunk_fused_off = false;
    for (dbm_model* m : c->models)
      if (m->type == 0 && !m->is_view) m->packed_dirty = true;  // the trunk's weight streams were not maintained meanwhile
    fprintf(stderr, "libdbm: persistent trunk kernels re-armed (iteration %ld)\n", g_step_serial);
  }
  if (c->dev_err && *(volatile int*)c->dev_err) {
    const bool dp = c->comm_active();
    dbm_handle_persistent_timeout(c);
    if (dp)
      throw DbmError(8, "a persistent kernel gave up waiting for a neighbouring workgroup in a data-parallel run: this rank skipped "
                        "optimizer updates the other ranks may have applied -- the replicas are no longer identical; abort the job");
    throw DbmError(7, "a persistent kernel gave up waiting for a neighbouring workgroup: " + std::to_string(c->timeout_skipped[0]) +
                          " discriminator / " + std::to_string(c->timeout_skipped[1]) +
                          " generator updates queued since were skipped (dbm_timeout_info); nothing of this call was enqueued -- "
                          "re-issue it (the layer-by-layer trunk path is active for a while)");
  }
}

#define DBM_API_END                                   \
  return 0;                                           \
  }                                                   \
  catch (const DbmError& e) {                         \
    g_last_error = e.what();                          \
    if (_ectx) _ectx->err = e.what();                 \
    return e.code;                                    \
  }                                                   \
  catch (const std::exception& e) {                   \
    g_last_error = e.what();                          \
    if (_ectx) _ectx->err = e.what();                 \
    return 3;                                         \
  }

// metrics finalisation -----------------------------------------------------------------------------
__global__ void gen_metrics_kernel(const float* part, int N, const float* adv, float* out, float nhw, float npool, float nwin,
                                   float cw, float aw, float tw, float sw) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float sums[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n = 0; n < N; ++n)  // per-tile partial sums of gen_loss_kernel, in tile order
    for (int k = 0; k < 4; ++k) sums[k] += part[4 * n + k];
  const float content = sums[0] / nhw;
  const float topo = sums[1] / npool;
  const float ssim = sums[2] / nwin;
  const float mse = sums[3] / nhw;
  out[0] = ((cw * content + aw * adv[0]) + tw * topo) + sw * (1.f - ssim);
  out[1] = 20.f * log10f(4294967296.f / sqrtf(mse));  // psnr, data_range = 2**32 (srgan_train.py:906-928)
  out[2] = ssim;
}

static void gaussian9(float* w, double sigma) {
  double g[9], s = 0;
  for (int i = 0; i < 9; ++i) { g[i] = exp(-((i - 4) * (i - 4)) / (2.0 * sigma * sigma)); s += g[i]; }
  for (int i = 0; i < 9; ++i) w[i] = (float)(g[i] / s);
}

// staging helpers for the host-pointer forms ---------------------------------------------------------
static const float* stage_in(dbm_ctx* c, int slot, const float* host, size_t n, int flags) {
  if (flags & DBM_DEVICE_PTRS) return host;
  if (!host) return nullptr;
  c->stage[slot].ensure(n);
  DBM_HIP(hipMemcpyAsync(c->stage[slot].p, host, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return c->stage[slot].p;
}
static float* stage_out(dbm_ctx* c, int slot, float* host, size_t n, int flags) {
  if (flags & DBM_DEVICE_PTRS) return host;
  if (!host) return nullptr;
  c->stage[slot].ensure(n);
  return c->stage[slot].p;
}
static void finish_out(dbm_ctx* c, int slot, float* host, size_t n, int flags) {
  if ((flags & DBM_DEVICE_PTRS) || !host) return;
  DBM_HIP(hipMemcpyAsync(host, c->stage[slot].p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
}
// End of a host-synchronising entry point (host pointers: the results are about to be read by the caller).  The stream is
// drained; if a persistent kernel gave up meanwhile, the results of THIS call are void as well: the condition is handled
// here (like at a step entry) and the call returns status 7 / 8 -- re-issue it, it then runs on the layer-by-layer kernels.
// Calls on device pointers only enqueue: their callers observe the condition with dbm_check_timeout after synchronising.
static void finish_sync(dbm_ctx* c, int flags) {
  if (flags & DBM_DEVICE_PTRS) return;
  DBM_HIP(hipStreamSynchronize(c->stream));
  if (c->dev_err && *(volatile int*)c->dev_err) {
    const bool dp = c->comm_active();
    dbm_handle_persistent_timeout(c);
    if (dp)
      throw DbmError(8, "a persistent kernel gave up waiting for a neighbouring workgroup in a data-parallel run: the replicas are "
                        "no longer identical; abort the job");
    throw DbmError(7, "a persistent kernel gave up waiting for a neighbouring workgroup: the results of this call are void -- "
                      "re-issue it (the layer-by-layer trunk path is active for a while)");
  }
}

extern "C" {

const char* dbm_last_error(dbm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int dbm_init(int hip_device, dbm_ctx** out) {
  DBM_API_BEGIN(nullptr)
  DBM_CHECK(out != nullptr, "dbm_init: out is NULL");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    throw DbmError(4, "dbm_init: no HIP device visible -- libdbm has no CPU fallback (the product path requires an MI355X)");
  DBM_CHECK(hip_device >= 0 && hip_device < count, "dbm_init: bad device index");
  DBM_HIP(hipSetDevice(hip_device));
  hipDeviceProp_t prop;
  DBM_HIP(hipGetDeviceProperties(&prop, hip_device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    throw DbmError(4, std::string("dbm_init: libdbm is built for gfx950 only, device is ") + prop.gcnArchName);
  dbm_ctx* c = new dbm_ctx();
  c->device = hip_device;
  {  // persistent trunk kernels: three workgroups per image, every workgroup of a launch resident at once (one per CU)
    int imgs = (prop.multiProcessorCount / 3) & ~7;
    c->trunk_imgs = imgs > 64 ? 64 : (imgs < 8 ? 8 : imgs);
  }
  DBM_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  {  // the side stream only carries filler work (weight gradients): lowest priority, so that the latency-bound
     // main chain gets CUs first whenever both have workgroups ready
    int least = 0, greatest = 0;
    DBM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    DBM_HIP(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, least));
  }
  // (round 5, VERDICT r4 1(c): chain[0] -- the discriminator's fake-batch backward pass -- at the lowest stream priority measured 7.90 / 7.94 ms
  //  against 7.92 / 7.92: no effect; both backward passes on ONE stream: 8.44 ms.  Not kept: profiles/r5/README.md)
  // (and chain[1] -- the generator's stream in dbm_train_iteration -- at the HIGHEST priority: 7.93-7.97 against 7.94: no effect either)
  for (auto& st : c->chain) DBM_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (auto& e : c->ev_fork) DBM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  DBM_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  DBM_HIP(hipHostMalloc((void**)&c->dev_err, sizeof(int), hipHostMallocMapped));
  *c->dev_err = 0;
  DBM_HIP(hipHostGetDevicePointer((void**)&c->dev_err_d, c->dev_err, 0));
  DBM_HIP(hipMalloc((void**)&c->dev_err_flag, 256));
  DBM_HIP(hipMemset(c->dev_err_flag, 0, 256));
  DBM_HIP(hipMalloc((void**)&c->zeros, 256));
  DBM_HIP(hipMemset(c->zeros, 0, 256));
  float w[9];
  gaussian9(w, 1.5);
  DBM_HIP(hipMalloc((void**)&c->ssim_win[0], sizeof(w)));
  DBM_HIP(hipMemcpy(c->ssim_win[0], w, sizeof(w), hipMemcpyHostToDevice));
  for (int i = 0; i < 9; ++i) w[i] = (float)(1.0 / 9.0);
  DBM_HIP(hipMalloc((void**)&c->ssim_win[1], sizeof(w)));
  DBM_HIP(hipMemcpy(c->ssim_win[1], w, sizeof(w), hipMemcpyHostToDevice));
  DBM_HIP(hipDeviceSynchronize());
  *out = c;
  DBM_API_END
}

int dbm_shutdown(dbm_ctx* ctx) {
  DBM_API_BEGIN(nullptr)
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ctx->zeros);
  if (ctx->dev_err) (void)hipHostFree(ctx->dev_err);
  (void)hipFree(ctx->dev_err_flag);
  (void)hipFree(ctx->ssim_win[0]);
  (void)hipFree(ctx->ssim_win[1]);
  ctx->comm_destroy();
  for (auto& e : ctx->ev_timer)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_iter)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_persist) (void)hipEventDestroy(ctx->ev_persist);
  if (ctx->ev_comm) (void)hipEventDestroy(ctx->ev_comm);
  if (ctx->ev_comm_done) (void)hipEventDestroy(ctx->ev_comm_done);
  for (auto& e : ctx->comm_ev_pool) (void)hipEventDestroy(e);
  ctx->loss_tmp.release();
  ctx->deferred.pending = false;   // (a pass nobody waited for: its metrics row dies with the context)
  ctx->deferred.fakes.release(); ctx->deferred.scratch.release(); ctx->deferred.logits.release();
  if (ctx->deferred.ev_ready) (void)hipEventDestroy(ctx->deferred.ev_ready);
  for (auto& b : ctx->stage) b.release();
  (void)hipStreamSynchronize(ctx->side);
  (void)hipStreamDestroy(ctx->side);
  for (auto& st : ctx->chain) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (auto& e : ctx->ev_fork) (void)hipEventDestroy(e);
  (void)hipEventDestroy(ctx->ev_join);
  (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  DBM_API_END
}

int dbm_set_stream(dbm_ctx* ctx, void* hip_stream) {
  DBM_API_BEGIN(ctx)  // (a pending iteration tail has been joined into the OLD stream by the line above: drain it)
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  DBM_API_END
}

int dbm_synchronize(dbm_ctx* ctx) {
  DBM_API_BEGIN(ctx)
  DBM_HIP(hipStreamSynchronize(ctx->side));
  for (auto& st : ctx->chain) DBM_HIP(hipStreamSynchronize(st));
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  DBM_API_END
}

int dbm_set_deterministic(dbm_ctx* ctx, int on) {
  DBM_API_BEGIN(ctx)
  g_wgrad_deterministic = on != 0;
  DBM_API_END
}

int dbm_set_sync_batch_stats(dbm_ctx* ctx, int world, void (*allreduce_sum)(void* user, float* dev, int n), void* user) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(world >= 1, "dbm_set_sync_batch_stats: world must be >= 1");
  DBM_CHECK(world == 1 || allreduce_sum != nullptr || (ctx->nccl_comm != nullptr && ctx->comm_world == world),
            "dbm_set_sync_batch_stats: world > 1 needs a hook or the native communicator (dbm_comm_init) of the same world");
  ctx->sync_world = world;
  ctx->sync_fn = world > 1 ? allreduce_sum : nullptr;
  ctx->sync_user = user;
  DBM_API_END
}

// ---- gradient exchange (comm.hip) ----
int dbm_comm_unique_id(void* out128) {
  DBM_API_BEGIN(nullptr)
  DBM_CHECK(out128 != nullptr, "dbm_comm_unique_id: out is NULL");
  dbm_comm_unique_id_impl(out128);
  DBM_API_END
}
int dbm_comm_init(dbm_ctx* ctx, int rank, int world, const void* id128) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(id128 != nullptr, "dbm_comm_init: id is NULL");
  ctx->comm_init(rank, world, id128);
  DBM_API_END
}
int dbm_comm_set_hook(dbm_ctx* ctx, int rank, int world,
                      void (*allreduce_sum)(void* user, float* dev, size_t n, void* hip_stream), void* user) {
  DBM_API_BEGIN(ctx)
  ctx->comm_set_hook(rank, world, allreduce_sum, user);
  DBM_API_END
}
int dbm_comm_destroy(dbm_ctx* ctx) {
  DBM_API_BEGIN(ctx)
  ctx->comm_destroy();
  DBM_API_END
}
int dbm_comm_broadcast(dbm_ctx* ctx, float* dev, size_t nfloats, int root) {
  DBM_API_BEGIN(ctx)
  ctx->comm_broadcast(dev, nfloats, root, ctx->stream);
  DBM_API_END
}
int dbm_comm_allreduce(dbm_ctx* ctx, float* dev, size_t nfloats) {
  DBM_API_BEGIN(ctx)
  ctx->comm_allreduce(&dev, &nfloats, 1, ctx->stream);
  DBM_API_END
}
int dbm_comm_stats(dbm_ctx* ctx, int* world, size_t* bytes, size_t* calls, int reset) {
  DBM_API_BEGIN(ctx)
  if (world) *world = ctx->comm_active() ? ctx->comm_world : 1;
  if (bytes) *bytes = ctx->comm_bytes;
  if (calls) *calls = ctx->comm_calls;
  if (reset) ctx->comm_bytes = ctx->comm_calls = 0;
  DBM_API_END
}
int dbm_allreduce_grads(dbm_model* m, double* grad_scale) {
  DBM_API_BEGIN(m->ctx)
  float* p = m->grads;
  size_t n = m->nparam;
  m->ctx->comm_allreduce(&p, &n, 1, m->ctx->stream);
  if (grad_scale) *grad_scale = m->ctx->comm_active() ? 1.0 / m->ctx->comm_world : 1.0;
  DBM_API_END
}

int dbm_profile_begin(dbm_ctx* ctx) {
  DBM_API_BEGIN(ctx)
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  double junk[12];
  g_profiler.collect(junk, 4);
  g_profiler.enabled = true;
  DBM_API_END
}

int dbm_profile_begin_serial(dbm_ctx* ctx) {
  DBM_API_BEGIN(ctx)
  DBM_HIP(hipDeviceSynchronize());
  double junk[12];
  g_profiler.collect(junk, 4);
  g_profiler.enabled = true;
  g_profiler.serial = true;
  DBM_API_END
}

int dbm_profile_end(dbm_ctx* ctx, double out[8]) {
  DBM_API_BEGIN(ctx)
  g_profiler.enabled = false;
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  double all[12];
  g_profiler.collect(all, 4);
  for (int i = 0; i < 6; ++i) out[i] = all[i];
  out[6] = out[7] = 0.0;
  DBM_API_END
}

int dbm_profile_end_ex(dbm_ctx* ctx, double* out, int nfam) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(out != nullptr && nfam >= 1 && nfam <= 5, "dbm_profile_end_ex: nfam must be 1..5");
  g_profiler.enabled = false;
  g_profiler.serial = false;
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  g_profiler.collect(out, nfam);
  DBM_API_END
}

int dbm_profile_end_records(dbm_ctx* ctx, char* buf, size_t cap, size_t* len) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(len != nullptr && (buf != nullptr || cap == 0), "dbm_profile_end_records: null argument");
  g_profiler.enabled = false;
  g_profiler.serial = false;
  DBM_HIP(hipDeviceSynchronize());
  static thread_local std::string pending;  // (a too-small buffer keeps the text for the retry)
  if (pending.empty()) pending = g_profiler.dump_records();
  *len = pending.size();
  if (pending.size() + 1 <= cap) {
    memcpy(buf, pending.c_str(), pending.size() + 1);
    pending.clear();
  }
  DBM_API_END
}

int dbm_memcpy2d_d2d(dbm_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
                     size_t height) {
  DBM_API_BEGIN(ctx)
  ctx->data_epoch++;  // caller-visible device memory changes: retained generator forwards keyed on it go stale
  if (width_bytes && height)
    DBM_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, hipMemcpyDeviceToDevice, ctx->stream));
  DBM_API_END
}

int dbm_gather_rows(dbm_ctx* ctx, void* dst, const void* src, const int* idx_host, int n, size_t row_bytes) {
  DBM_API_BEGIN(ctx)
  ctx->data_epoch++;  // caller-visible device memory changes: retained generator forwards keyed on it go stale
  DBM_CHECK(n >= 0 && row_bytes % 4 == 0, "dbm_gather_rows: rows must be whole float32 elements");
  if (n && row_bytes) {
    DBM_CHECK(idx_host != nullptr, "dbm_gather_rows: idx is NULL");
    ctx->stage[7].ensure((size_t)n);  // (floats: 4 bytes each, like the int indices)
    DBM_HIP(hipMemcpyAsync(ctx->stage[7].p, idx_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    launch_gather_rows(src, dst, (const int*)ctx->stage[7].p, n, row_bytes, ctx->stream);
  }
  DBM_API_END
}

int dbm_fill_f32(dbm_ctx* ctx, float* dst, size_t n, float value) {
  DBM_API_BEGIN(ctx)
  ctx->data_epoch++;  // caller-visible device memory changes: retained generator forwards keyed on it go stale
  if (n) launch_fill(dst, (long)n, value, ctx->stream);
  DBM_API_END
}

int dbm_clip_min_f32(dbm_ctx* ctx, float* dst, size_t n, float lo) {
  DBM_API_BEGIN(ctx)
  ctx->data_epoch++;  // caller-visible device memory changes: retained generator forwards keyed on it go stale
  if (n) launch_clip_min(dst, (long)n, lo, ctx->stream);
  DBM_API_END
}

int dbm_debug_inject_timeout(dbm_ctx* ctx) {
  DBM_API_BEGIN(nullptr)  // (not observed by this call itself)
  DBM_CHECK(ctx != nullptr, "dbm_debug_inject_timeout: ctx is NULL");
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  const int one = 1;
  DBM_HIP(hipMemcpy(ctx->dev_err_flag, &one, sizeof(int), hipMemcpyHostToDevice));
  *(volatile int*)ctx->dev_err = 1;
  DBM_API_END
}

__global__ void debug_raise_timeout_kernel(int* err_host, int* err_dev) {
  *err_dev = 1;
  *err_host = 1;
}
int dbm_debug_inject_timeout_async(dbm_ctx* ctx) {
  DBM_API_BEGIN(nullptr)
  DBM_CHECK(ctx != nullptr, "dbm_debug_inject_timeout_async: ctx is NULL");
  hipLaunchKernelGGL(debug_raise_timeout_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->dev_err_d, ctx->dev_err_flag);
  DBM_HIP(hipGetLastError());
  DBM_API_END
}

int dbm_check_timeout(dbm_ctx* ctx) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(ctx != nullptr, "dbm_check_timeout: ctx is NULL");
  dbm_step_entry(ctx, false);
  DBM_API_END
}

int dbm_timeout_info(dbm_ctx* ctx, long* events, int* d_updates_skipped, int* g_updates_skipped, int* persistent_off) {
  DBM_API_BEGIN(nullptr)
  DBM_CHECK(ctx != nullptr, "dbm_timeout_info: ctx is NULL");
  if (events) *events = ctx->timeout_events;
  if (d_updates_skipped) *d_updates_skipped = ctx->timeout_skipped[0];
  if (g_updates_skipped) *g_updates_skipped = ctx->timeout_skipped[1];
  if (persistent_off) *persistent_off = g_trunk_fused_off ? 1 : 0;
  DBM_API_END
}

int dbm_timer(dbm_ctx* ctx, int op, double* ms) {
  DBM_API_BEGIN(ctx)
  for (auto& e : ctx->ev_timer)
    if (!e) DBM_HIP(hipEventCreate(&e));
  if (op == 0) {
    DBM_HIP(hipEventRecord(ctx->ev_timer[0], ctx->stream));
  } else if (op == 1) {
    DBM_HIP(hipEventRecord(ctx->ev_timer[1], ctx->stream));
  } else {
    DBM_CHECK(ms != nullptr, "dbm_timer: ms is NULL");
    DBM_HIP(hipEventSynchronize(ctx->ev_timer[1]));
    float t = 0.f;
    DBM_HIP(hipEventElapsedTime(&t, ctx->ev_timer[0], ctx->ev_timer[1]));
    *ms = (double)t;
  }
  DBM_API_END
}

int dbm_phase_marks(dbm_ctx* ctx, int enable, char* out, int cap) {
  DBM_API_BEGIN(ctx)
  if (enable) {
    (void)g_profiler.dump_marks();
    g_profiler.marks_enabled = true;
  } else {
    g_profiler.marks_enabled = false;
    DBM_HIP(hipStreamSynchronize(ctx->side));
    DBM_HIP(hipStreamSynchronize(ctx->stream));
    const std::string txt = g_profiler.dump_marks();
    if (out && cap > 0) {
      const size_t n = std::min(txt.size(), (size_t)cap - 1);
      memcpy(out, txt.data(), n);
      out[n] = 0;
    }
  }
  DBM_API_END
}

int dbm_malloc(dbm_ctx* ctx, size_t bytes, void** dptr) {
  DBM_API_BEGIN(ctx)
  ctx->data_epoch++;  // caller-visible device memory changes: retained generator forwards keyed on it go stale
  // 128-byte guards on both sides (see DevBuf): kernels may read one word outside a tensor
  char* base = nullptr;
  DBM_HIP(hipMalloc((void**)&base, bytes + 256));
  DBM_HIP(hipMemset(base, 0, bytes + 256));
  DBM_HIP(hipDeviceSynchronize());
  *dptr = base + 128;
  DBM_API_END
}
int dbm_free(dbm_ctx* ctx, void* dptr) {
  DBM_API_BEGIN(ctx)
  ctx->data_epoch++;  // caller-visible device memory changes: retained generator forwards keyed on it go stale
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  DBM_HIP(hipFree(dptr ? (char*)dptr - 128 : nullptr));
  DBM_API_END
}
int dbm_memcpy_h2d(dbm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  DBM_API_BEGIN(ctx)
  ctx->data_epoch++;  // caller-visible device memory changes: retained generator forwards keyed on it go stale
  DBM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  DBM_API_END
}
int dbm_memcpy_d2h(dbm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  DBM_API_BEGIN(ctx)
  DBM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  DBM_API_END
}

// ---- models ----
int dbm_gen_create(dbm_ctx* ctx, int n, float rs, int oc, dbm_model** out) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(ctx && out, "dbm_gen_create: NULL argument");
  *out = new Generator(ctx, n, rs, oc);
  DBM_API_END
}
int dbm_disc_create(dbm_ctx* ctx, dbm_model** out) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(ctx && out, "dbm_disc_create: NULL argument");
  *out = new Discriminator(ctx);
  DBM_API_END
}
int dbm_model_destroy(dbm_model* m) {
  DBM_API_BEGIN(m ? m->ctx : nullptr)
  if (m) {
    (void)hipStreamSynchronize(m->ctx->stream);
    delete m;
  }
  DBM_API_END
}
int dbm_model_num_tensors(dbm_model* m, int* n) {
  DBM_API_BEGIN(m->ctx)
  *n = (int)m->tensors.size();
  DBM_API_END
}
int dbm_model_tensor_info(dbm_model* m, int i, const char** key, int* ndim, int64_t shape[4], int* kind) {
  DBM_API_BEGIN(m->ctx)
  DBM_CHECK(i >= 0 && i < (int)m->tensors.size(), "tensor index out of range");
  const Tensor& t = m->tensors[i];
  *key = t.key.c_str();
  *ndim = t.ndim;
  for (int k = 0; k < 4; ++k) shape[k] = t.shape[k];
  *kind = t.kind;
  DBM_API_END
}
static float* tensor_ptr(dbm_model* m, const Tensor& t, bool grad) {
  if (t.kind == DBM_KIND_PARAM) return (grad ? m->grads : m->params) + t.off;
  DBM_CHECK(!grad, "persistent values have no gradient");
  return m->pers + t.off;
}
int dbm_model_set_tensor(dbm_model* m, const char* key, const float* host, size_t n) {
  DBM_API_BEGIN(m->ctx)
  const Tensor& t = m->tensors[m->tid(key)];
  DBM_CHECK(n == t.n, std::string("size mismatch for ") + key);
  DBM_HIP(hipMemcpyAsync(tensor_ptr(m, t, false), host, n * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
  DBM_HIP(hipStreamSynchronize(m->ctx->stream));
  m->packed_dirty = true;
  m->param_version++;
  DBM_API_END
}
int dbm_model_get_tensor(dbm_model* m, const char* key, float* host, size_t n) {
  DBM_API_BEGIN(m->ctx)
  const Tensor& t = m->tensors[m->tid(key)];
  DBM_CHECK(n == t.n, std::string("size mismatch for ") + key);
  DBM_HIP(hipMemcpyAsync(host, tensor_ptr(m, t, false), n * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
  DBM_HIP(hipStreamSynchronize(m->ctx->stream));
  DBM_API_END
}
int dbm_model_get_grad(dbm_model* m, const char* key, float* host, size_t n) {
  DBM_API_BEGIN(m->ctx)
  const Tensor& t = m->tensors[m->tid(key)];
  DBM_CHECK(n == t.n, std::string("size mismatch for ") + key);
  DBM_HIP(hipMemcpyAsync(host, tensor_ptr(m, t, true), n * sizeof(float), hipMemcpyDeviceToHost, m->ctx->stream));
  DBM_HIP(hipStreamSynchronize(m->ctx->stream));
  DBM_API_END
}
int dbm_model_count_params(dbm_model* m, int64_t* n) {
  DBM_API_BEGIN(m->ctx)
  *n = (int64_t)m->nparam;
  DBM_API_END
}
int dbm_model_cleargrads(dbm_model* m) {
  DBM_API_BEGIN(m->ctx)
  DBM_HIP(hipMemsetAsync(m->grads, 0, m->nparam * sizeof(float), m->ctx->stream));
  mark_grads_cleared(m);  // (a view shares its owner's arena: both marks go)
  DBM_API_END
}
int dbm_model_param_arena(dbm_model* m, void** dptr, size_t* n) {
  DBM_API_BEGIN(m->ctx)
  *dptr = m->params;
  *n = m->nparam;
  DBM_API_END
}
int dbm_model_grad_arena(dbm_model* m, void** dptr, size_t* n) {
  DBM_API_BEGIN(m->ctx)
  *dptr = m->grads;
  *n = m->nparam;
  DBM_API_END
}
int dbm_model_params_changed(dbm_model* m) {
  DBM_API_BEGIN(m->ctx)
  m->packed_dirty = true;
  m->param_version++;
  DBM_API_END
}

// ---- forward / backward ----
int dbm_gen_forward(dbm_model* gm, int N, int H, int W, const float* x, const float* w1, const float* w2,
                    const float* w3, float* y, int flags) {
  DBM_API_BEGIN(gm->ctx)
  DBM_CHECK(gm->type == 0, "dbm_gen_forward: not a generator");
  DBM_CHECK(N >= 1 && x && w1 && w2 && w3 && y, "dbm_gen_forward: bad arguments");
  Generator* g = static_cast<Generator*>(gm);
  const bool keep = flags & DBM_KEEP_GRAPH;
  const size_t n = (size_t)N, hw = (size_t)H * W, P4 = 16 * (size_t)(H - 2) * (W - 2) * (size_t)g->out_ch;
  struct Bf16Scope {  // DBM_BF16: the convolution descriptors of this call point at the bf16 weight images
    Generator* g;
    Bf16Scope(Generator* gen, bool on) : g(on ? gen : nullptr) {
      if (g) { g->ensure_packed_bf16(); g->use_bf16 = true; }
    }
    ~Bf16Scope() { if (g) g->use_bf16 = false; }
  };
  DBM_CHECK(!((flags & DBM_BF16) && keep), "dbm_gen_forward: DBM_BF16 is an inference mode (no DBM_KEEP_GRAPH)");
  Bf16Scope bf16(g, (flags & DBM_BF16) != 0);
  if (flags & DBM_DEVICE_PTRS) {
    g->forward(N, H, W, x, w1, w2, w3, y, keep);
  } else {
    g->ensure_ws(N, H, W, keep);
    hipStream_t s = g->ctx->stream;
    DBM_HIP(hipMemcpyAsync(g->in_x.p, x, n * hw * 4, hipMemcpyHostToDevice, s));
    DBM_HIP(hipMemcpyAsync(g->in_w1.p, w1, n * 100 * hw * 4, hipMemcpyHostToDevice, s));
    DBM_HIP(hipMemcpyAsync(g->in_w2.p, w2, n * 8 * hw * 4, hipMemcpyHostToDevice, s));
    DBM_HIP(hipMemcpyAsync(g->in_w3.p, w3, n * hw * 4, hipMemcpyHostToDevice, s));
    g->forward(N, H, W, g->in_x.p, g->in_w1.p, g->in_w2.p, g->in_w3.p, g->yout.p, keep);
    DBM_HIP(hipMemcpyAsync(y, g->yout.p, n * P4 * 4, hipMemcpyDeviceToHost, s));
    finish_sync(g->ctx, flags);   // (observes a persistent-kernel time-out: status 7, the results are void)
  }
  DBM_API_END
}

int dbm_gen_backward(dbm_model* gm, const float* gy, int flags) {
  DBM_API_BEGIN(gm->ctx)
  DBM_CHECK(gm->type == 0, "dbm_gen_backward: not a generator");
  Generator* g = static_cast<Generator*>(gm);
  if (flags & DBM_DEVICE_PTRS) {
    g->backward(gy);
  } else {
    DBM_CHECK(g->have_graph, "generator backward without a retained forward (DBM_KEEP_GRAPH)");
    const size_t cnt = (size_t)g->wsN * 16 * (g->wsH - 2) * (g->wsW - 2);
    DBM_HIP(hipMemcpyAsync(g->g_y.p, gy, cnt * 4, hipMemcpyHostToDevice, g->ctx->stream));
    g->backward(g->g_y.p);
    finish_sync(g->ctx, flags);
  }
  DBM_API_END
}

int dbm_disc_forward(dbm_model* dm, int N, int H, int W, const float* img, float* logits, int flags, int slot) {
  DBM_API_BEGIN(dm->ctx)
  DBM_CHECK(dm->type == 1, "dbm_disc_forward: not a discriminator");
  Discriminator* d = static_cast<Discriminator*>(dm);
  dbm_ctx* c = d->ctx;
  const size_t n = (size_t)N;
  // the image is retained by pointer for the backward pass: host input goes to a per-slot staging buffer
  DBM_CHECK(slot == 0 || slot == 1, "dbm_disc_forward: cache slot must be 0 or 1");
  const float* dimg = stage_in(c, 4 + slot, img, n * H * W, flags);
  float* dlog = stage_out(c, 6 + slot, logits, n, flags);
  d->forward(N, H, W, dimg, dlog, flags & DBM_BN_TRAIN, flags & DBM_KEEP_GRAPH, slot);
  finish_out(c, 6 + slot, logits, n, flags);
  finish_sync(c, flags);
  DBM_API_END
}

int dbm_disc_backward(dbm_model* dm, int slot, const float* glogits, int flags) {
  DBM_API_BEGIN(dm->ctx)
  DBM_CHECK(dm->type == 1, "dbm_disc_backward: not a discriminator");
  Discriminator* d = static_cast<Discriminator*>(dm);
  DBM_CHECK(slot == 0 || slot == 1, "bad slot");
  const float* g = stage_in(d->ctx, 0, glogits, (size_t)d->cache[slot].N, flags);
  d->backward(slot, g);
  finish_sync(d->ctx, flags);
  DBM_API_END
}

// ---- losses ----
static int discriminator_loss_impl(dbm_ctx* ctx, const float* real, const float* fake, int N, int t_rf, int t_fr,
                                   const int* t_rf_arr, const int* t_fr_arr, float* out2, float* g_real, float* g_fake,
                                   int flags) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(N >= 1 && real && fake && out2, "dbm_discriminator_loss: bad arguments");
  const float* dr = stage_in(ctx, 0, real, N, flags);
  const float* df = stage_in(ctx, 1, fake, N, flags);
  float* dout = stage_out(ctx, 2, out2, 2, flags);
  float* dgr = stage_out(ctx, 3, g_real, N, flags);
  float* dgf = stage_out(ctx, 4, g_fake, N, flags);
  const int* dt1 = (const int*)stage_in(ctx, 5, (const float*)t_rf_arr, N, flags);  // (int32: the same four bytes per element)
  const int* dt2 = (const int*)stage_in(ctx, 6, (const float*)t_fr_arr, N, flags);
  launch_ragan_loss(dr, df, N, t_rf, t_fr, dout, dgr, dgf, ctx->stream, dt1, dt2);
  finish_out(ctx, 2, out2, 2, flags);
  finish_out(ctx, 3, g_real, N, flags);
  finish_out(ctx, 4, g_fake, N, flags);
  finish_sync(ctx, flags);
  DBM_API_END
}
int dbm_discriminator_loss(dbm_ctx* ctx, const float* real, const float* fake, int N, int t_rf, int t_fr, float* out2,
                           float* g_real, float* g_fake, int flags) {
  return discriminator_loss_impl(ctx, real, fake, N, t_rf, t_fr, nullptr, nullptr, out2, g_real, g_fake, flags);
}
int dbm_discriminator_loss_t(dbm_ctx* ctx, const float* real, const float* fake, int N, const int* t_rf, const int* t_fr,
                             float* out2, float* g_real, float* g_fake, int flags) {
  if (!t_rf || !t_fr) {
    g_last_error = "dbm_discriminator_loss_t: both target arrays are required";
    if (ctx) ctx->err = g_last_error;
    return 1;
  }
  return discriminator_loss_impl(ctx, real, fake, N, 1, 0, t_rf, t_fr, out2, g_real, g_fake, flags);
}

// device-side generator loss; all pointers are device pointers.  out3: [g_loss, psnr, ssim].
// Two halves, because the adversarial term is DETACHED from the generator (srgan_train.py:1229-1237): the gradient gy only
// needs the content / topographic / SSIM terms (gen_loss_terms), the discriminator's logits only enter the loss VALUE
// (gen_loss_finish) -- the G-step runs its discriminator forward beside the generator's backward pass.
static void gen_loss_terms(dbm_ctx* ctx, const float* y, const float* t, const float* X, int N, int H, int W, const float w[4],
                           int win, float* gy) {
  DBM_CHECK(win == 0 || win == 1, "ssim_window must be 0 (gaussian) or 1 (uniform)");
  hipStream_t s = ctx->stream;
  ctx->loss_tmp.ensure(16 + 5 * (size_t)N);
  float* sums = ctx->loss_tmp.p + 16 + N;  // 4 N per-tile partial sums
  DBM_HIP(hipMemsetAsync(ctx->loss_tmp.p, 0, 16 * sizeof(float), s));
  launch_gen_loss(y, t, X, N, H, W, w[0], w[2], w[3], ctx->ssim_win[win], sums, gy, s);
}
// adversarial term: calculate_discriminator_loss(real=ones, fake=D(fake) detached, targets swapped) (:874-879, :1233-1237)
static void gen_loss_adv(dbm_ctx* ctx, const float* real_logits, const float* fake_logits, int N, int t_rf, int t_fr,
                         const int* t_rf_arr = nullptr, const int* t_fr_arr = nullptr, float* base = nullptr) {
  hipStream_t s = ctx->stream;
  if (!base) base = ctx->loss_tmp.p;   // (the deferred eval-mode pass brings its own copy of the scratch)
  float* adv = base + 8;    // [8..9]
  float* ones = base + 16;  // N
  if (!real_logits) {
    launch_fill(ones, N, 1.f, s);
    real_logits = ones;
  }
  launch_ragan_loss(real_logits, fake_logits, N, t_rf, t_fr, adv, nullptr, nullptr, s, t_rf_arr, t_fr_arr);
}
static void gen_loss_finish(dbm_ctx* ctx, int N, int H, int W, const float w[4], float* out3, const float* base = nullptr) {
  hipStream_t s = ctx->stream;
  if (!base) base = ctx->loss_tmp.p;
  const float* adv = base + 8;
  const float* sums = base + 16 + N;
  const float nhw = (float)N * H * W, npool = (float)N * (H / 4) * (W / 4), nwin = (float)N * (H - 8) * (W - 8);
  hipLaunchKernelGGL(gen_metrics_kernel, dim3(1), dim3(64), 0, s, sums, N, adv, out3, nhw, npool, nwin, w[0], w[1], w[2], w[3]);
  DBM_HIP(hipGetLastError());
}
static void gen_loss_device(dbm_ctx* ctx, const float* y, const float* t, const float* X, const float* real_logits,
                            const float* fake_logits, int N, int H, int W, const float w[4], int t_rf, int t_fr,
                            int win, float* out3, float* gy, const int* t_rf_arr = nullptr, const int* t_fr_arr = nullptr) {
  gen_loss_terms(ctx, y, t, X, N, H, W, w, win, gy);
  gen_loss_adv(ctx, real_logits, fake_logits, N, t_rf, t_fr, t_rf_arr, t_fr_arr);
  gen_loss_finish(ctx, N, H, W, w, out3);
}

static int generator_loss_impl(dbm_ctx* ctx, const float* y_pred, const float* y_true, const float* x, const float* real_logits,
                               const float* fake_logits, int N, int H, int W, const float weights[4], int t_rf, int t_fr,
                               const int* t_rf_arr, const int* t_fr_arr, int ssim_window, float* out3, float* gy, int flags) {
  DBM_API_BEGIN(ctx)
  const size_t n = (size_t)N, hw = (size_t)H * W, xhw = (size_t)(H / 4 + 2) * (W / 4 + 2);
  const float* dy = stage_in(ctx, 0, y_pred, n * hw, flags);
  const float* dt = stage_in(ctx, 1, y_true, n * hw, flags);
  const float* dx = stage_in(ctx, 2, x, n * xhw, flags);
  const float* dl = stage_in(ctx, 3, fake_logits, n, flags);
  const float* dr = stage_in(ctx, 6, real_logits, n, flags);
  float* dout = stage_out(ctx, 4, out3, 3, flags);
  float* dgy = stage_out(ctx, 5, gy, n * hw, flags);
  // (slot 7 holds both target arrays back to back: the eight staging slots are otherwise taken)
  const int* dt1 = nullptr;
  const int* dt2 = nullptr;
  if (t_rf_arr && t_fr_arr) {
    if (flags & DBM_DEVICE_PTRS) {
      dt1 = t_rf_arr; dt2 = t_fr_arr;
    } else {
      ctx->stage[7].ensure(2 * n);
      DBM_HIP(hipMemcpyAsync(ctx->stage[7].p, t_rf_arr, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
      DBM_HIP(hipMemcpyAsync(ctx->stage[7].p + n, t_fr_arr, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
      dt1 = (const int*)ctx->stage[7].p; dt2 = dt1 + n;
    }
  }
  gen_loss_device(ctx, dy, dt, dx, dr, dl, N, H, W, weights, t_rf, t_fr, ssim_window, dout, dgy, dt1, dt2);
  finish_out(ctx, 4, out3, 3, flags);
  finish_out(ctx, 5, gy, n * hw, flags);
  finish_sync(ctx, flags);
  DBM_API_END
}
int dbm_generator_loss(dbm_ctx* ctx, const float* y_pred, const float* y_true, const float* x, const float* real_logits,
                       const float* fake_logits, int N, int H, int W, const float weights[4], int t_rf, int t_fr,
                       int ssim_window, float* out3, float* gy, int flags) {
  return generator_loss_impl(ctx, y_pred, y_true, x, real_logits, fake_logits, N, H, W, weights, t_rf, t_fr, nullptr, nullptr,
                             ssim_window, out3, gy, flags);
}
int dbm_generator_loss_t(dbm_ctx* ctx, const float* y_pred, const float* y_true, const float* x, const float* real_logits,
                         const float* fake_logits, int N, int H, int W, const float weights[4], const int* t_rf, const int* t_fr,
                         int ssim_window, float* out3, float* gy, int flags) {
  if (!t_rf || !t_fr) {
    g_last_error = "dbm_generator_loss_t: both target arrays are required";
    if (ctx) ctx->err = g_last_error;
    return 1;
  }
  return generator_loss_impl(ctx, y_pred, y_true, x, real_logits, fake_logits, N, H, W, weights, 0, 1, t_rf, t_fr, ssim_window,
                             out3, gy, flags);
}

__global__ void psnr_finish_kernel(const float* part, int blocks, float* out, float n, float range) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < blocks; ++i) s += part[i];
  out[0] = 20.f * log10f(range / sqrtf(s / n));
}
__global__ void ssim_finish_kernel(const float* part, int N, float nwin, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += part[4 * n + 2];
  out[0] = s / nwin;
}

int dbm_psnr(dbm_ctx* ctx, const float* y_pred, const float* y_true, size_t n, double data_range, float* out, int flags) {
  DBM_API_BEGIN(ctx)
  const float* a = stage_in(ctx, 0, y_pred, n, flags);
  const float* b = stage_in(ctx, 1, y_true, n, flags);
  const int blocks = sqdiff_blocks((long)n);
  ctx->loss_tmp.ensure(32 + (size_t)blocks);
  float* acc = ctx->loss_tmp.p;
  launch_sqdiff(a, b, (long)n, acc + 32, ctx->stream);
  hipLaunchKernelGGL(psnr_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, acc + 32, blocks, acc + 1, (float)n, (float)data_range);
  if (flags & DBM_DEVICE_PTRS) {
    DBM_HIP(hipMemcpyAsync(out, acc + 1, sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    DBM_HIP(hipMemcpyAsync(out, acc + 1, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    DBM_HIP(hipStreamSynchronize(ctx->stream));
  }
  DBM_API_END
}

int dbm_ssim(dbm_ctx* ctx, const float* y_pred, const float* y_true, int N, int H, int W, int ssim_window, float* out,
             int flags) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(ssim_window == 0 || ssim_window == 1, "ssim_window must be 0 (gaussian) or 1 (uniform)");
  const size_t cnt = (size_t)N * H * W;
  const float* a = stage_in(ctx, 0, y_pred, cnt, flags);
  const float* b = stage_in(ctx, 1, y_true, cnt, flags);
  float* dout = stage_out(ctx, 2, out, 1, flags);
  ctx->loss_tmp.ensure(32 + 4 * (size_t)N);
  float* sums = ctx->loss_tmp.p + 32;
  launch_gen_loss(a, b, nullptr, N, H, W, 0.f, 0.f, 0.f, ctx->ssim_win[ssim_window], sums, nullptr, ctx->stream);
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, sums, N, (float)N * (H - 8) * (W - 8), dout);
  finish_out(ctx, 2, out, 1, flags);
  finish_sync(ctx, flags);
  DBM_API_END
}

int dbm_ssim_ex(dbm_ctx* ctx, const float* y_pred, const float* y_true, int N, int H, int W, int window_size, int stride,
                int ssim_window, float* out, int flags) {
  if (window_size == 9 && stride == 1) return dbm_ssim(ctx, y_pred, y_true, N, H, W, ssim_window, out, flags);
  DBM_API_BEGIN(ctx)
  DBM_CHECK(ssim_window == 0 || ssim_window == 1, "ssim_window must be 0 (gaussian) or 1 (uniform)");
  DBM_CHECK(window_size >= 1 && window_size <= 64 && stride >= 1, "dbm_ssim_ex: window_size in [1, 64], stride >= 1");
  DBM_CHECK(H >= window_size && W >= window_size, "dbm_ssim_ex: the images are smaller than the window");
  const size_t cnt = (size_t)N * H * W;
  const float* a = stage_in(ctx, 0, y_pred, cnt, flags);
  const float* b = stage_in(ctx, 1, y_true, cnt, flags);
  float* dout = stage_out(ctx, 2, out, 1, flags);
  ctx->loss_tmp.ensure(32 + 4 * (size_t)N);
  float* sums = ctx->loss_tmp.p + 32;
  launch_ssim_general(a, b, N, H, W, window_size, stride, ssim_window, sums, ctx->stream);
  const float nwin = (float)N * (float)((H - window_size) / stride + 1) * (float)((W - window_size) / stride + 1);
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, sums, N, nwin, dout);
  finish_out(ctx, 2, out, 1, flags);
  finish_sync(ctx, flags);
  DBM_API_END
}

// ---- optimizer ----
int dbm_adam_setup(dbm_model* m, double alpha, double beta1, double beta2, double eps) {
  DBM_API_BEGIN(m->ctx)
  m->alpha = alpha; m->beta1 = beta1; m->beta2 = beta2; m->eps = eps;
  m->adam_t = 0;
  m->adam_ready = true;
  DBM_HIP(hipMemsetAsync(m->adam_m, 0, m->nparam * sizeof(float), m->ctx->stream));
  DBM_HIP(hipMemsetAsync(m->adam_v, 0, m->nparam * sizeof(float), m->ctx->stream));
  DBM_API_END
}

static void adam_update_impl(dbm_model* m, double grad_scale) {
  DBM_CHECK(m->adam_ready, "dbm_adam_update before dbm_adam_setup");
  m->adam_t += 1;
  const double fix1 = 1.0 - std::pow(m->beta1, (double)m->adam_t);
  const double fix2 = 1.0 - std::pow(m->beta2, (double)m->adam_t);
  const double alpha_t = m->alpha * std::sqrt(fix2) / fix1;  // AdamRule.alpha_t
  DBM_MARK(m->ctx->stream, m->type == 0 ? "G:optimizer_begin" : "D:optimizer_begin");
  // (the sticky timeout flag is sampled ONCE per launch into the model's gate word: the update is all or nothing)
  launch_adam(m->params, m->grads, m->adam_m, m->adam_v, (long)m->nparam, (float)alpha_t, (float)(1.0 - m->beta1),
              (float)(1.0 - m->beta2), (float)m->eps, (float)grad_scale, m->ctx->stream, m->ctx->dev_err_flag,
              m->d_adam_skipped);
  DBM_MARK(m->ctx->stream, m->type == 0 ? "G:optimizer" : "D:optimizer");
  m->packed_dirty = true;
  m->param_version++;
  // The generator's packed weight images and trunk weight streams are rebuilt right behind its update: the caller is
  // about to fetch this iteration's metrics (a host round trip during which the GPU would idle), and the next forward
  // finds them ready instead of starting with 0.13 ms of repacking.  (The discriminator's repack already hides under
  // the generator passes that precede its next use.)
  if (m->type == 0) m->ensure_packed();
}

int dbm_adam_update(dbm_model* m, double grad_scale) {
  DBM_API_BEGIN(m->ctx)
  try {
    dbm_step_entry(m->ctx, false);
  } catch (const DbmError& e) {
    // The event lies BEFORE this update: the gradients in the arena come from a void pass, and the handler has just cleared
    // the sticky flag -- an update re-issued now would apply them.  Status 9: nothing was applied; redo forward + backward.
    if (e.code == 7)
      throw DbmError(9, "dbm_adam_update: a persistent kernel gave up during the pass that produced these gradients -- they are "
                        "void and NOTHING was applied; repeat the forward and backward pass (or the step call), then update");
    throw;
  }
  // ... or the event was observed EARLIER, by a host-synchronising call between the void backward pass and this update (an
  // eval-mode forward that answered 7 and was repeated): the flag is clean again, the arena is not
  if (m->grads_void)
    throw DbmError(9, "dbm_adam_update: a persistent-kernel time-out was handled since this model's gradients were last cleared -- "
                      "they may come from a void pass and NOTHING was applied; cleargrads, repeat forward and backward, then update");
  adam_update_impl(m, grad_scale);
  DBM_API_END
}

// ---- fused steps ----
int dbm_discriminator_step(dbm_model* gm, dbm_model* dm, int N, int H, int W, const float* X, const float* W1,
                           const float* W2, const float* W3, const float* Y, int train, float* metrics) {
  DBM_API_BEGIN(gm->ctx)
  DBM_CHECK(gm->type == 0 && dm->type == 1, "dbm_discriminator_step: (generator, discriminator) expected");
  Generator* g = static_cast<Generator*>(gm);
  Discriminator* d = static_cast<Discriminator*>(dm);
  dbm_ctx* c = g->ctx;
  dbm_step_entry(c, (train & 1) != 0);
  hipStream_t s = c->stream;
  const int H4 = 4 * (H - 2), W4 = 4 * (W - 2);
  const bool share = (train & 2) != 0;  // opt-in: keep this forward's graph for the G-step of the same iteration
  // opt-in (the trainer's pattern: the G-step of the same minibatch follows): that step's generator forward -- same
  // weights, same inputs, its own workspace -- is enqueued on separate streams behind this step's forward, so that it
  // runs underneath the discriminator passes instead of after them.  Nothing is skipped; the order of independent
  // work changes.
  const bool prefetch = (train & 4) != 0 && !share;
  // bit 8: the caller runs collectives on a stream of its own (RCCL): the prefetched forward then stays on ONE library
  // stream so that, with the caller's two, no more than four are ever busy (see Generator::twin)
  const bool narrow = (train & 8) != 0 && !g->trunk_fused_ok(H - 2, W - 2);  // (a fused pass is one stream anyway)
  // data-parallel run with a communicator on the context (dbm_comm_init / dbm_comm_set_hook): the gradient buckets are
  // summed over ranks inside this call, overlapped with the backward passes; bit 4 (16) leaves the exchange to the caller
  struct CommScope {
    dbm_ctx* c;
    CommScope(dbm_ctx* ctx, bool on) : c(ctx) { c->comm_in_step = on; }
    ~CommScope() { c->comm_in_step = false; }
  } comm_scope(c, (train & 1) && !(train & 16) && c->comm_active());
  train &= 1;
  // whatever an earlier call retained for a following G-step is void now (n_critic > 1 loops, refilled arrays)
  g->graph_version = -1;
  if (g->twin) g->twin->graph_version = -1;
  g->ensure_ws(N, H, W, share && train);
  d->g_out.ensure(4 * (size_t)N);
  float* lr = d->g_out.p;
  float* lf = lr + N;
  float* gr = lf + N;
  float* gf = gr + N;
  // D(real) does not depend on the generator: it runs on the side stream underneath the (latency-bound) generator
  // forward.  D(fake) starts only after it has finished, so the two BatchNorm running-average updates keep the
  // reference's order (real, then fake: srgan_train.py:1145-1146).
  DBM_MARK(s, "D:begin");
  // sync_batch_stats: the statistics hooks enqueue their collectives on the main stream, so every discriminator pass
  // stays there (no side / chain streams for them)
  const bool sync = c->sync_stats() && train;
  c->fork_to_side(0);
  {
    hipStream_t main_stream = c->stream;
    if (!sync) c->stream = c->side;
    try {
      d->forward(N, H4, W4, Y, lr, train, train, 0);  // real batch (:1145)
    } catch (...) {
      c->stream = main_stream;
      throw;
    }
    c->stream = main_stream;
  }
  // fake images under enable_backprop=False (:1131-1137)
  g->forward(N, H, W, X, W1, W2, W3, g->yout.p, share && train);
  g->graph_version = g->param_version;
  g->graph_epoch = c->data_epoch;
  g->graph_in[0] = X; g->graph_in[1] = W1; g->graph_in[2] = W2; g->graph_in[3] = W3;
  DBM_MARK(s, "D:generator_forward");
  if (prefetch && train) {
    Generator* t = g->get_twin();
    t->ensure_ws(N, H, W, true);
    if (!g->ev_prefetch) DBM_HIP(hipEventCreateWithFlags(&g->ev_prefetch, hipEventDisableTiming));
    t->max_split = narrow ? 1 : 2;
    hipStream_t pf = narrow ? c->chain[0] : c->chain[1];
    c->fork(s, pf, 6);  // weights packed, inputs final, and not before this step's own forward is done
    c->stream = pf;
    try {
      t->forward(N, H, W, X, W1, W2, W3, t->yout.p, true);
    } catch (...) {
      c->stream = s;
      throw;
    }
    DBM_HIP(hipEventRecord(g->ev_prefetch, pf));
    t->max_split = 2;
    c->stream = s;
    t->graph_version = g->param_version;
    t->graph_epoch = c->data_epoch;
    t->graph_in[0] = X; t->graph_in[1] = W1; t->graph_in[2] = W2; t->graph_in[3] = W3;
  }
  c->join_side();
  d->forward(N, H4, W4, g->yout.p, lf, train, train, 1);   // fake batch (:1146) -- separate BatchNorm statistics
  if (sync) {  // relativistic means over the global batch (:995-1004)
    float* sb = c->sync_buf.p + 3 * 512;
    launch_ragan_sync_sums(lr, lf, N, sb, s);
    c->allreduce(sb, 2);
    launch_ragan_sync_loss(lr, lf, N, c->sync_world, 1, 0, sb, metrics, s);
    c->allreduce(sb + 2, 2);
    launch_ragan_sync_grad(lr, lf, N, c->sync_world, 1, 0, sb, gr, gf, s);
  } else {
    launch_ragan_loss(lr, lf, N, 1, 0, metrics, train ? gr : nullptr, train ? gf : nullptr, s);
  }
  DBM_MARK(s, "D:disc_forward_fake+loss");
  if (train) {
    DBM_HIP(hipMemsetAsync(d->grads, 0, d->nparam * sizeof(float), s));  // cleargrads (:1162)
    mark_grads_cleared(d);
    // d_loss.backward() (:1163): the real- and the fake-batch graphs are independent (gradients are accumulated
    // with atomics), so the fake batch's pass runs on a second stream; both hand their weight gradients to the side stream
    // (while a prefetched generator forward owns chain[0] / chain[1], both passes stay on the main stream)
    // (with the persistent trunk kernel the prefetched forward occupies ONE stream, chain[1]: chain[0] is free again)
    const bool twin_one_stream = g->trunk_fused_ok(H - 2, W - 2);
    const bool two_streams = (!(prefetch && train) || twin_one_stream) && !sync;
    d->ensure_packed_bwd(s);   // (before the fork: both passes read the data-gradient images)
    if (two_streams) {
      c->fork(s, c->chain[0], 7);
      c->stream = c->chain[0];
    }
    d->merge_slots = true;  // one weight-gradient launch per layer group for both graphs
    d->comm_sent_lo = d->comm_sent_hi = 0;
    try {
      // The pass on the MAIN stream (real batch) is enqueued first: the main stream is the one the step's tail waits
      // for (measured: D-step 5.72 -> 5.39 ms against enqueueing the fake batch's pass first); whichever pass is enqueued
      // second launches the merged weight-gradient groups behind both passes' events.
      static const bool fake_first = DBM_TUNE_GETENV("DBWD_ORDER") && atoi(DBM_TUNE_GETENV("DBWD_ORDER")) == 0;
      hipStream_t other = c->stream;  // chain[0] when two streams are used, else the main stream
      if (fake_first) {
        d->merge_launcher = 0;
        d->backward(1, gf, false);
        c->stream = s;
        d->backward(0, gr, false);
      } else {
        d->merge_launcher = 1;
        c->stream = s;
        d->backward(0, gr, false);
        c->stream = other;
        d->backward(1, gf, false);
        c->stream = s;
      }
    } catch (...) {
      c->stream = s;
      d->merge_slots = false;
      throw;
    }
    d->merge_slots = false;
    DBM_MARK(s, "D:disc_backward_real_chain");
    if (two_streams) c->fork(c->chain[0], s, 7);
    DBM_MARK(s, "D:disc_backward_fake_chain_joined");
    c->join_side();
    DBM_MARK(s, "D:weight_gradients_joined");
    if (c->comm_in_step) {  // what launch_group has not sent yet: [0, lo) and [hi, nparam) in one fused group
      float* p[2] = {d->grads, d->grads + d->comm_sent_hi};
      size_t n[2] = {d->comm_sent_lo, d->nparam - d->comm_sent_hi};
      if (d->comm_sent_hi == 0) { n[0] = d->nparam; n[1] = 0; }
      c->comm_bucket(p, n, n[1] ? 2 : 1, s);
      c->comm_join(s);
      DBM_MARK(s, "D:gradients_exchanged");
    }
  }
  DBM_API_END
}

int dbm_generator_step(dbm_model* gm, dbm_model* dm, int N, int H, int W, const float* X, const float* W1,
                       const float* W2, const float* W3, const float* Y, const float weights[4], int ssim_window,
                       int train, float* metrics) {
  DBM_API_BEGIN(gm->ctx)
  DBM_CHECK(gm->type == 0 && dm->type == 1, "dbm_generator_step: (generator, discriminator) expected");
  Generator* g = static_cast<Generator*>(gm);
  Discriminator* d = static_cast<Discriminator*>(dm);
  dbm_ctx* c = g->ctx;
  dbm_step_entry(c, false);
  hipStream_t s = c->stream;
  const int H4 = 4 * (H - 2), W4 = 4 * (W - 2);
  const bool share = (train & 2) != 0;
  // bit 2 (4): the caller asserts that the five arrays are the ones, unchanged, the preceding dbm_discriminator_step saw:
  // only then may the forward that step prefetched be consumed.  (Pointers, shapes, the parameter version and the
  // library's own record of writes to device memory are checked on top; writes by anybody else are invisible here.)
  const bool use_prefetched = (train & 4) != 0;
  struct CommScope {
    dbm_ctx* c;
    CommScope(dbm_ctx* ctx, bool on) : c(ctx) { c->comm_in_step = on; }
    ~CommScope() { c->comm_in_step = false; }
  } comm_scope(c, (train & 1) && !(train & 16) && c->comm_active());
  train &= 1;
  // Opt-in: the generator and its inputs are unchanged since the D-step of this iteration, so that step's forward
  // (the same numbers: it is the retained form of the pass) is reused instead of recomputed.  Off by default: the
  // reference runs it twice.
  const bool reuse = share && train && g->have_graph && g->wsTrain && g->graph_version == g->param_version &&
                     g->graph_epoch == c->data_epoch && g->wsN == N && g->wsH == H && g->wsW == W && g->graph_in[0] == X && g->graph_in[1] == W1 &&
                     g->graph_in[2] == W2 && g->graph_in[3] == W3;
  DBM_MARK(s, "G:begin");
  Generator* t = g->twin;
  const bool prefetched = train && use_prefetched && t && t->have_graph && t->wsTrain && t->graph_version == g->param_version &&
                          t->graph_epoch == c->data_epoch && t->wsN == N &&
                          t->wsH == H && t->wsW == W && t->graph_in[0] == X && t->graph_in[1] == W1 && t->graph_in[2] == W2 &&
                          t->graph_in[3] == W3;
  Generator* gg = prefetched ? t : g;  // the workspace that holds this step's graph
  const bool pack_aside = d->packed_dirty && !reuse && !prefetched && !train;
  if (pack_aside) {  // the discriminator's weight images (stale since its Adam step) are rebuilt under the generator forward
    c->fork_to_side(5);
    d->ensure_packed(c->side);
  }
  // Training: the discriminator's eval-mode forward on the fakes only feeds the loss VALUE (detached, :1229-1237), so it
  // runs on chain[1] -- weight repack included -- beside the generator's backward pass; main joins it for the metrics.
  const bool overlap_d = train && !reuse;
  if (prefetched) {
    if (!overlap_d) d->ensure_packed();
    DBM_HIP(hipStreamWaitEvent(s, g->ev_prefetch, 0));
    t->graph_version = -1;  // consumed
  } else if (!reuse) {
    g->ensure_ws(N, H, W, train != 0);
    g->forward(N, H, W, X, W1, W2, W3, g->yout.p, train != 0);  // (:1222-1227)
  }
  DBM_MARK(s, "G:generator_forward");
  if (pack_aside) c->join_side();
  d->g_out.ensure(4 * (size_t)N);
  float* lf = d->g_out.p;
  if (overlap_d) {
    gen_loss_terms(c, gg->yout.p, Y, X, N, H4, W4, weights, ssim_window, gg->g_y.p);
    hipStream_t aux = c->chain[1];
    c->fork(s, aux, 11);  // fakes, the discriminator's updated weights and the cleared loss scratch are final on `s`
    c->stream = aux;
    try {
      d->forward(N, H4, W4, gg->yout.p, lf, false, false, 1);  // eval-mode BatchNorm, detached (:1228-1229)
      gen_loss_adv(c, nullptr, lf, N, 0, 1);
    } catch (...) {
      c->stream = s;
      throw;
    }
    c->stream = s;
    DBM_MARK(s, "G:disc_forward+loss");
    DBM_HIP(hipMemsetAsync(g->grads, 0, g->nparam * sizeof(float), s));  // cleargrads (:1255)
    mark_grads_cleared(g);
    gg->grads_cleared = true;  // (the memset above: two-slice weight gradients may fold with atomics, bit for bit)
    gg->backward(gg->g_y.p);                                             // g_loss.backward() (:1256)
    gg->grads_cleared = false;
    c->fork(aux, s, 12);  // (chain[1] also carried the gradient exchange of a data-parallel run)
    gen_loss_finish(c, N, H4, W4, weights, metrics + 2);
    DBM_MARK(s, "G:generator_backward_joined");
  } else {
    d->forward(N, H4, W4, gg->yout.p, lf, false, false, 1);  // eval-mode BatchNorm, detached (:1228-1229)
    gen_loss_device(c, gg->yout.p, Y, X, nullptr, lf, N, H4, W4, weights, 0, 1, ssim_window, metrics + 2,
                    train ? gg->g_y.p : nullptr);
    DBM_MARK(s, "G:disc_forward+loss");
    if (train) {
      DBM_HIP(hipMemsetAsync(g->grads, 0, g->nparam * sizeof(float), s));  // cleargrads (:1255)
      mark_grads_cleared(g);
      gg->grads_cleared = true;
      gg->backward(gg->g_y.p);                                             // g_loss.backward() (:1256)
      gg->grads_cleared = false;
      c->comm_join(s);
      DBM_MARK(s, "G:generator_backward_joined");
    }
  }
  DBM_API_END
}

// One training iteration (srgan_train.py:1286-1309: train_eval_discriminator, then train_eval_generator, both optimizer
// updates included) as ONE call, scheduled as a whole.  Nothing is skipped or reordered numerically: the discriminator
// is updated from its own gradients first, the generator's gradients come from its own retained forward, the adversarial
// term of g_loss is evaluated with the UPDATED discriminator -- but the generator's backward pass does not depend on
// anything the D-step computes (the adversarial term is detached, :1228-1229), so it is enqueued behind the prefetched
// forward on chain[1] and runs underneath the discriminator's backward passes and weight gradients, whose chains of
// small kernels leave most of the chip idle.  Bitwise the same result as the two step calls + two dbm_adam_update calls.
// Data-parallel run (a communicator on the context, round 3): the same schedule; the gradient buckets of BOTH models are
// summed over ranks on chain[0] -- the discriminator's big bucket (conv_layer6..9) behind the fake-batch backward pass
// that stream carries, its remainder after the join, then the generator's buckets (tail, trunk groups, input block) as
// its backward pass on chain[1] finishes them -- and both Adam launches take 1 / world.  Same collectives in the same order
// as the two step calls, hence the same numbers bit for bit.
static void run_deferred_eval(dbm_ctx* c, hipStream_t on) {
  dbm_ctx::DeferredEval& q = c->deferred;
  if (!q.pending) return;
  q.pending = false;
  struct Restore { dbm_ctx* c; hipStream_t s; ~Restore() { c->stream = s; } } restore{c, c->stream};
  DBM_HIP(hipStreamWaitEvent(on, q.ev_ready, 0));
  c->stream = on;
  // (the discriminator's forward weight images were rebuilt behind its update; coefficients: prepare_eval_coeffs(2) at the snapshot)
  q.d->forward(q.N, q.H4, q.W4, q.fakes.p, q.logits.p, false, false, 2, true);
  gen_loss_adv(c, nullptr, q.logits.p, q.N, 0, 1, nullptr, nullptr, q.scratch.p);
  gen_loss_finish(c, q.N, q.H4, q.W4, q.w, q.out3, q.scratch.p);
}

int dbm_train_iteration(dbm_model* gm, dbm_model* dm, int N, int H, int W, const float* X, const float* W1, const float* W2,
                        const float* W3, const float* Y, const float weights[4], int ssim_window, int flags, float* metrics) {
  DBM_API_BEGIN_NOFLUSH(gm->ctx)
  DBM_CHECK(gm->type == 0 && dm->type == 1, "dbm_train_iteration: (generator, discriminator) expected");
  Generator* g = static_cast<Generator*>(gm);
  Discriminator* d = static_cast<Discriminator*>(dm);
  dbm_ctx* c = g->ctx;
  dbm_step_entry(c, true);
  DBM_CHECK(g->adam_ready && d->adam_ready, "dbm_train_iteration: both optimizers must be set up (dbm_adam_setup)");
  DBM_CHECK(!c->sync_stats(), "dbm_train_iteration: sync_batch_stats runs use the two step calls (their statistics collectives "
                              "are enqueued on the main stream between the layers)");
  DBM_CHECK(ssim_window == 0 || ssim_window == 1, "ssim_window must be 0 (gaussian) or 1 (uniform)");
  const bool one_fwd = (flags & DBM_ONE_GEN_FORWARD) != 0;  // the retained forward also supplies the D-step's fakes
  hipStream_t s = c->stream;
  hipStream_t pf = c->chain[1];
  const int H4 = 4 * (H - 2), W4 = 4 * (W - 2);
  const bool dp = c->comm_active();
  const double gscale = dp ? 1.0 / c->comm_world : 1.0;
  g->graph_version = -1;
  if (g->twin) g->twin->graph_version = -1;
  g->ensure_ws(N, H, W, false);
  d->g_out.ensure(5 * (size_t)N);
  float* lr = d->g_out.p;
  float* lf = lr + N;
  float* gr = lf + N;
  float* gf = gr + N;
  float* lf_eval = gf + N;  // logits of the G-step's eval-mode pass (lf / gf are still being read by the backward passes)
  if (!c->ev_iter[0]) for (auto& e : c->ev_iter) DBM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (!g->ev_prefetch) DBM_HIP(hipEventCreateWithFlags(&g->ev_prefetch, hipEventDisableTiming));
  // (The iteration's tail -- the discriminator's weight repack, the G-step's detached eval-mode discriminator pass and the
  // metrics -- runs on the main stream.  Round 3 tried three alternatives (profiles/design_history_r1-r3.md): raising chain[1]'s
  // stream priority lost its A/B; moving the tail to chain[0] (DBM_ITER_TAIL) was NEVER validly measured -- DBM_API_BEGIN's
  // join cleared tail_pending, so both arms ran the same schedule -- and the path was deleted without a re-measurement;
  // forking the twin's forward early is measured again in round 5: DBM_ITER_EARLY_TWIN below.)
  struct Scope {  // every helper below enqueues on ctx->stream / reads the exchange switches: restore them whatever happens
    dbm_ctx* c; hipStream_t s; Discriminator* d; Generator* t = nullptr;
    ~Scope() {
      c->stream = s; c->comm_in_step = false; c->comm_defer = false; c->comm_stream = nullptr; c->comm_pending.clear();
      d->merge_slots = false; d->borrow_images = false;
      if (t) { t->grads_cleared = false; t->use_aux = true; t->max_split = 2; t->wgrad_inline = false; t->csr_early = false; t->csr_prebuilt = false; }
    }
  } scope{c, s, d};
  c->comm_in_step = dp;
  c->comm_stream = dp ? c->chain[0] : nullptr;
  // (tuning switch, libdbm_measure.so only: DBM_DISC_BORROW=0 private image copies again.  profiles/r6/ab_disc_launch_trims.txt, three
  //  alternations on one box, medians: all three trims of round 6 on 7.668 ms per step; the head as two launches per pass 7.695; image copies
  //  7.681; the D-step's cleargrads at the head of the side stream instead of between the loss and the backward passes 7.668 against 7.656
  //  WITHOUT it -- that one lost and is gone again: 77 MB of fills in front of D(real)'s forward cost more than 41 MB behind the loss.)
  static const int borrow_env = DBM_TUNE_GETENV("DISC_BORROW") ? atoi(DBM_TUNE_GETENV("DISC_BORROW")) : 1;
  d->borrow_images = borrow_env != 0;   // (Y and the generator's output buffers outlive this call; both backward passes run inside it)
  DBM_MARK(s, "D:begin");
  // DBM_ITER_DEFER_EVAL=1 (round 6; default 0): the G-step's detached eval-mode discriminator pass of THIS iteration is snapshotted at its
  // end and enqueued by the next library call (dbm_ctx::DeferredEval); 0: inside this call, behind the discriminator's update.
  // MEASURED (profiles/r6/ab_defer_eval.txt, two alternations on one box): inside the call 7.616-7.618 ms per step; deferred to the side
  // stream beside D(fake)'s forward 7.68-7.69, in front of D(real)'s forward or first thing on the main stream 7.84-7.85, first thing on
  // chain[0] 7.68-7.72 (7.66-7.69 with the weight repack left to D(real)'s forward).  Removing the pass is worth 0.32 ms (round 5's
  // ablation), but every other place it can go costs more than the place it has: behind the update it runs on the main stream while that
  // stream only waits for the generator's backward pass, squeezed into whatever the chain and the trunk's weight gradients leave; anywhere
  // in the next iteration it competes with a forward pass on the critical path.  The mechanism stays (bitwise the same metrics:
  // tests/test_gpu_round5.py) for callers whose next call is NOT a training iteration -- the pass then costs nothing until it is read.
  // Where a pending pass of the PREVIOUS iteration goes (tuning switch, libdbm_measure.so only): 0 main stream, first thing; 1 side
  // stream, in front of D(real)'s forward; 2 (default) side stream, behind D(real)'s forward and the weight-image rebuilds -- beside
  // D(fake)'s forward, in the shadow of the retained trunk forward; 3 chain[0], first thing.
  static const int defer_env = getenv("DBM_ITER_DEFER_EVAL") ? atoi(getenv("DBM_ITER_DEFER_EVAL")) : 0;
  static const int defer_at = DBM_TUNE_GETENV("ITER_DEFER_AT") ? atoi(DBM_TUNE_GETENV("ITER_DEFER_AT")) : 2;
  static const int defer_pack = DBM_TUNE_GETENV("ITER_DEFER_PACK") ? atoi(DBM_TUNE_GETENV("ITER_DEFER_PACK")) : 1;
  if (c->deferred.pending && (defer_at == 0 || c->deferred.d != d)) run_deferred_eval(c, s);
  if (c->deferred.pending && defer_at == 3) run_deferred_eval(c, c->chain[0]);
  // (libdbm_measure.so only; results are then wrong -- what a part of the iteration costs INSIDE it: 1 = no discriminator work at all
  //  (forwards, backward passes, weight gradients, update, repack, eval-mode pass), 2 = no trunk weight-gradient launch (generator.hip),
  //  4 = no weight gradients of the generator's tail, 8 = no eval-mode discriminator pass)
  static const int iter_abl = DBM_MEASURE_ENV("ITER_ABL");
  const bool no_d = (iter_abl & 1) != 0;
  // ---- D(real) forward on the side stream, underneath the generator forward (:1145) ----
  g->ensure_packed();   // (normally done behind the previous update; never on the side stream: the forward below reads these images)
  c->fork_to_side(0);
  c->stream = c->side;
  // cleargrads of the G-step (:1255), early: nothing reads or writes the generator's gradient arena between the previous update and this
  // iteration's backward pass, and 35 MB of fill would otherwise sit between the loss and the backward pass on the critical path
  DBM_HIP(hipMemsetAsync(g->grads, 0, g->nparam * sizeof(float), c->side));
  if (c->deferred.pending && defer_at == 1) run_deferred_eval(c, c->side);
  if (!no_d) d->forward(N, H4, W4, Y, lr, true, true, 0);
  c->stream = s;
  // (The G-step's own forward goes to chain[1] behind the first forward; one_fwd: it is the only forward, forked here.
  //  Round 4 measured releasing only its INPUT BLOCK early -- beside the first forward's tail, the trunk launch still behind it:
  //  8.073 against 8.027 ms, two alternations on one box: not kept.)
  // DBM_ITER_EARLY_TWIN (round 5; persistent trunk path, single GPU): where the G-step's own forward is released.
  //   0: behind the WHOLE first forward (its full-resolution tail included);
  //   1: behind the first forward's TRUNK launch -- the persistent launches are serialised anyway (persist_begin), and the first
  //      forward's tail (a serial chain of seven short kernels) then runs beside the second trunk's 192 workgroups instead of alone;
  //   2: FIRST -- the retained forward's trunk precedes the D-step's in the chain of persistent launches: the generator's own
  //      forward -> loss -> backward -> update path is the iteration's critical path, the D-step has slack behind it.
  // Nothing is skipped and no number changes: the two forwards read the same weights and inputs and write separate workspaces.
  static const int early_env = getenv("DBM_ITER_EARLY_TWIN") ? atoi(getenv("DBM_ITER_EARLY_TWIN")) : 0;
  const int early = (!one_fwd && !dp && g->trunk_fused_ok(H - 2, W - 2)) ? early_env : 0;
  if (one_fwd || early == 2) {
    g->ensure_packed();
    c->fork(s, pf, 6);
  }
  Generator* t = g->get_twin();
  scope.t = t;
  t->ensure_ws(N, H, W, true);
  // DBM_ITER_CSR_EARLY (default 1): the deformable layers' sampling lists (they depend on the offsets only) are built on chain[0]
  // behind the discriminator's fake-batch pass -- i.e. beside the generator's loss -- instead of on the backward pass's own path
  // (Generator::prebuild_csr; 7.69-7.70 against 7.77-7.78 ms.  On the side stream, which must then wait for the retained forward's
  // tail, the discriminator's weight gradients start late: 8.07)
  static const int csr_early_env = getenv("DBM_ITER_CSR_EARLY") ? atoi(getenv("DBM_ITER_CSR_EARLY")) : 1;
  auto twin_forward = [&]() {  // the G-step's own forward (:1222-1227), retained graph, second workspace, on chain[1]
    t->max_split = 1;
    t->csr_early = csr_early_env != 0 && !dp;   // (data-parallel: chain[0] also carries the gradient exchange -- the lists stay on the backward pass's own path)
    c->stream = pf;
    t->forward(N, H, W, X, W1, W2, W3, t->yout.p, true);
    DBM_HIP(hipEventRecord(g->ev_prefetch, pf));  // the twin's fakes are final (the G-step's eval-mode discriminator pass reads them)
    c->stream = s;
    t->max_split = 2;
  };
  if (early == 2) twin_forward();
  // ---- fakes under enable_backprop=False (:1131-1137) ----
  if (!one_fwd) {
    g->mark_trunk = early == 1;
    g->forward(N, H, W, X, W1, W2, W3, g->yout.p, false);
    g->mark_trunk = false;
  }
  DBM_MARK(s, "D:generator_forward");
  if (early == 1) DBM_HIP(hipStreamWaitEvent(pf, g->ev_trunk, 0));  // weights packed, inputs final, the first trunk launch enqueued
  else if (!one_fwd && early == 0) c->fork(s, pf, 6);
  if (early != 2) twin_forward();
  // ---- D(fake) forward, RaGAN loss, cleargrads (:1146-1162) ----
  c->join_side();
  // The data-gradient weight images of both models (stale since their updates; first read by this iteration's backward passes) are rebuilt
  // HERE on the side stream -- behind D(real)'s forward and behind the join, i.e. beside D(fake)'s forward -- instead of between the
  // discriminator's update and its eval-mode pass in the previous iteration.  (At the head of the side stream they sat in front of
  // D(real)'s forward, and the whole queue waits while the helper-form trunk holds every CU: D(real) started 0.09 ms later.)
  g->ensure_packed_bwd(c->side);
  DBM_HIP(hipEventRecord(c->ev_iter[2], c->side));   // (what the generator's backward pass on chain[1] waits for: cleargrads + images)
  d->ensure_packed_bwd(c->side);
  DBM_HIP(hipEventRecord(c->ev_iter[3], c->side));   // (what the discriminator's backward passes wait for)
  if (c->deferred.pending) run_deferred_eval(c, c->side);   // (defer_at == 2; the main stream joins the side stream before the update below)
  if (one_fwd) DBM_HIP(hipStreamWaitEvent(s, g->ev_prefetch, 0));  // (the fakes are the retained forward's, written on chain[1])
  if (!no_d) d->forward(N, H4, W4, one_fwd ? t->yout.p : g->yout.p, lf, true, true, 1);
  launch_ragan_loss(lr, lf, N, 1, 0, metrics, gr, gf, s);
  DBM_MARK(s, "D:disc_forward_fake+loss");
  DBM_HIP(hipMemsetAsync(d->grads, 0, d->nparam * sizeof(float), s));
  mark_grads_cleared(d);
  // ---- d_loss.backward() (:1163): real batch on the main stream, fake batch on chain[0], weight gradients on side ----
  DBM_HIP(hipStreamWaitEvent(s, c->ev_iter[3], 0));   // (before the fork: both passes read the data-gradient images)
  c->fork(s, c->chain[0], 7);
  d->merge_slots = true;
  d->merge_launcher = 1;
  d->comm_sent_lo = d->comm_sent_hi = 0;
  c->comm_defer = dp;  // (chain[0] is the exchange stream AND carries the fake-batch pass: its bucket goes out behind the pass)
  if (!no_d) d->backward(0, gr, false);
  c->stream = c->chain[0];
  if (!no_d) d->backward(1, gf, false);
  c->stream = s;
  d->merge_slots = false;
  c->comm_defer = false;
  if (dp) c->comm_flush();
  c->fork(c->chain[0], s, 7);
  if (csr_early_env && !dp) t->prebuild_csr(c->chain[0]);   // (behind the fake-batch pass, and behind the mark the main stream waits for)
  c->join_side();  // (the discriminator's weight gradients: everything on the side stream so far)
  DBM_MARK(s, "D:weight_gradients_joined");
  if (dp) {  // what launch_group has not sent yet: [0, lo) and [hi, nparam) in one fused group; then the optimizer's wait
    float* p[2] = {d->grads, d->grads + d->comm_sent_hi};
    size_t n[2] = {d->comm_sent_lo, d->nparam - d->comm_sent_hi};
    if (d->comm_sent_hi == 0) { n[0] = d->nparam; n[1] = 0; }
    c->comm_bucket(p, n, n[1] ? 2 : 1, s);
    c->comm_join(s);
    DBM_MARK(s, "D:gradients_exchanged");
  }
  // ---- the generator's loss terms and backward pass (:1248-1256) on chain[1], behind its forward ----
  c->stream = pf;
  DBM_MARK(pf, "G:retained_forward_done");
  gen_loss_terms(c, t->yout.p, Y, X, N, H4, W4, weights, ssim_window, t->g_y.p);
  DBM_MARK(pf, "G:loss_terms");
  DBM_HIP(hipEventRecord(c->ev_iter[0], pf));
  DBM_HIP(hipStreamWaitEvent(pf, c->ev_iter[2], 0));  // cleargrads (:1255): the fill at the head of the side stream (+ the weight images)
  mark_grads_cleared(g);
  t->grads_cleared = true;
  // (chain[0] carries the discriminator's fake-batch pass and the gradient exchange.  DBM_ITER_AUX=1, single GPU only: the offset-
  // gradient kernel of final_conv_layer2 goes there all the same, next to the input-gradient gather)
  static const int iter_aux = DBM_TUNE_GETENV("ITER_AUX") ? atoi(DBM_TUNE_GETENV("ITER_AUX")) : 0;
  t->use_aux = iter_aux && !dp;
  t->wgrad_inline = early == 2;   // (the side stream carries the discriminator's weight gradients: see Generator::wgrad_inline)
  t->backward(t->g_y.p);
  t->wgrad_inline = false;
  t->grads_cleared = false;
  t->use_aux = true;
  t->graph_version = -1;
  DBM_HIP(hipEventRecord(c->ev_iter[1], pf));
  c->stream = s;
  // ---- discriminator update (:1164), then the G-step's detached eval-mode discriminator pass (:1228-1237) ----
  if (!no_d) adam_update_impl(d, gscale);
  DBM_HIP(hipStreamWaitEvent(s, g->ev_prefetch, 0));  // the twin's fakes (written on chain[1]: nothing else orders this read)
  if (defer_env && !no_d && !(iter_abl & 8)) {
    // snapshot what the eval-mode pass reads; the pass itself goes out with the next library call (run_deferred_eval)
    dbm_ctx::DeferredEval& q = c->deferred;
    if (defer_pack) d->ensure_packed(s);   // (the forward weight images of the updated parameters: D(real)'s next forward needs them as well)
    d->prepare_eval_coeffs(2, s);
    const size_t nimg = (size_t)N * H4 * W4;
    q.fakes.ensure(nimg); q.scratch.ensure(16 + 5 * (size_t)N); q.logits.ensure((size_t)N);
    DBM_HIP(hipMemcpyAsync(q.fakes.p, t->yout.p, nimg * sizeof(float), hipMemcpyDeviceToDevice, s));
    DBM_HIP(hipStreamWaitEvent(s, c->ev_iter[0], 0));  // the loss terms' partial sums (chain[1])
    DBM_HIP(hipMemsetAsync(q.scratch.p, 0, 16 * sizeof(float), s));
    DBM_HIP(hipMemcpyAsync(q.scratch.p + 16 + N, c->loss_tmp.p + 16 + N, 4 * (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (!q.ev_ready) DBM_HIP(hipEventCreateWithFlags(&q.ev_ready, hipEventDisableTiming));
    DBM_HIP(hipEventRecord(q.ev_ready, s));
    q.d = d; q.N = N; q.H4 = H4; q.W4 = W4; q.out3 = metrics + 2;
    for (int k = 0; k < 4; ++k) q.w[k] = weights[k];
    q.pending = true;
  } else {
    if (!no_d && !(iter_abl & 8)) d->forward(N, H4, W4, t->yout.p, lf_eval, false, false, 1);  // (repacks the updated weights first)
    DBM_HIP(hipStreamWaitEvent(s, c->ev_iter[0], 0));  // the loss scratch was cleared on chain[1]
    gen_loss_adv(c, nullptr, lf_eval, N, 0, 1);
    gen_loss_finish(c, N, H4, W4, weights, metrics + 2);   // (the loss terms are final: not behind the backward pass's join -- it sat between the
                                                           //  last weight gradient and the update, 12 us on the iteration's critical path)
  }
  DBM_HIP(hipStreamWaitEvent(s, c->ev_iter[1], 0));  // generator backward (and its weight gradients) done
  if (dp) {
    c->comm_join(s);  // ... and its last bucket summed over ranks
    DBM_MARK(s, "G:gradients_exchanged");
  }
  DBM_MARK(s, "G:generator_backward_joined");
  adam_update_impl(g, gscale);  // (:1257)
  DBM_API_END
}

// ---- op-level entry points ----
static IgLayer make_temp_layer(dbm_ctx* ctx, const float* w, int O, int C, int k, int stride, int pad, dbm_model& holder) {
  // a throw-away single-layer "model" whose param arena aliases the caller's weights
  holder.ctx = ctx;
  holder.add_tensor("op/W", {O, C, k, k}, DBM_KIND_PARAM);
  holder.add_tensor("op/b", {O}, DBM_KIND_PARAM);
  holder.alloc_arenas();
  DBM_HIP(hipMemcpyAsync(holder.params, w, sizeof(float) * (size_t)O * C * k * k, hipMemcpyDeviceToDevice, ctx->stream));
  holder.add_iglayer("op", O, C, k, stride, pad, true);
  holder.ensure_packed();
  return holder.layers[0];
}

int dbm_op_conv2d(dbm_ctx* ctx, const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W,
                  int O, int k, int stride, int pad, int ups, int lrelu) {
  DBM_API_BEGIN(ctx)
  if (C % 32 != 0) {
    DBM_CHECK(!ups, "few-channel conv has no fused upsample");
    SmallConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.xsn = (long)C * H * W; d.Cin = C; d.Hin = H; d.Win = W; d.w = w; d.bias = b;
    d.OH = (H + 2 * pad - k) / stride + 1; d.OW = (W + 2 * pad - k) / stride + 1;
    d.y = y; d.ysn = (long)O * d.OH * d.OW; d.Cout = O; d.KH = d.KW = k; d.stride = stride; d.pad = pad; d.N = N;
    d.act = lrelu; d.slope = 0.2f;
    launch_smallcin_conv_fwd(d, ctx->stream);
  } else {
    dbm_model holder;
    IgLayer L = make_temp_layer(ctx, w, O, C, k, stride, pad, holder);
    if (b) DBM_HIP(hipMemcpyAsync(holder.P(L.bi), b, sizeof(float) * O, hipMemcpyDeviceToDevice, ctx->stream));
    const int Hl = H << ups, Wl = W << ups;
    const int OH = (Hl + 2 * pad - k) / stride + 1, OW = (Wl + 2 * pad - k) / stride + 1;
    ConvDesc d = holder.fwd_desc(L, x, (long)C * H * W, H, W, ups, y, (long)O * OH * OW, N);
    d.act = lrelu;
    launch_igemm_conv(d, ctx->stream);
    DBM_HIP(hipStreamSynchronize(ctx->stream));
  }
  DBM_API_END
}

int dbm_op_conv2d_backward(dbm_ctx* ctx, const float* x, const float* w, const float* gy, float* gx, float* gw,
                           float* gb, int N, int C, int H, int W, int O, int k, int stride, int pad, int ups) {
  DBM_API_BEGIN(ctx)
  const int Hl = H << ups, Wl = W << ups;
  const int OH = (Hl + 2 * pad - k) / stride + 1, OW = (Wl + 2 * pad - k) / stride + 1;
  if (C % 32 != 0) {
    DBM_CHECK(!ups && gx == nullptr, "few-channel conv: only the weight gradient exists on the hot path");
    SmallConvDesc q;
    memset(&q, 0, sizeof(q));
    q.x = x; q.xsn = (long)C * H * W; q.Cin = C; q.Hin = H; q.Win = W; q.Cout = O; q.OH = OH; q.OW = OW;
    q.KH = q.KW = k; q.stride = stride; q.pad = pad; q.N = N;
    launch_smallcin_conv_wgrad(q, gy, (long)O * OH * OW, gw, gb, ctx->stream);
  } else {
    DBM_CHECK(O % 32 == 0 || gx == nullptr, "op dgrad needs O % 32 == 0 (pad the gradient channels)");
    dbm_model holder;
    IgLayer L = make_temp_layer(ctx, w, O, C, k, stride, pad, holder);
    if (gw) {
      WgradDesc wd;
      memset(&wd, 0, sizeof(wd));
      wd.x = x; wd.xsn = (long)C * H * W; wd.xsc = H * W; wd.Cin = C; wd.Hin = H; wd.Win = W; wd.ups = ups;
      wd.dy = gy; wd.dysn = (long)O * OH * OW; wd.dysc = OH * OW; wd.Cout = O; wd.OH = OH; wd.OW = OW;
      wd.KH = wd.KW = k; wd.stride = stride; wd.pad = pad; wd.N = N; wd.scale = 1.f; wd.gW = gw; wd.gb = gb;
      launch_wgrad(wd, ctx->stream);
    }
    if (gx) {
      ConvDesc d;
      memset(&d, 0, sizeof(d));
      d.x = gy; d.xsn = (long)O * OH * OW; d.N = N;
      d.y = gx; d.ysn = (long)C * Hl * Wl; d.s1 = 1.f; d.s2 = 1.f;
      holder.ensure_packed_bwd();
      holder.run_dgrad(L, d, Hl, Wl);  // gradient w.r.t. the (upsampled) conv input
    }
    DBM_HIP(hipStreamSynchronize(ctx->stream));
  }
  DBM_API_END
}

// 3x3 / stride 1 / pad 1 convolution through the channels-last bf16 kernel of the sweep's trunk (conv_cl16.hip), for the parity
// tests: x (N, C, H, W) fp32 is rounded to bf16 NHWC, y = [lrelu](s1 * (conv + b) + r1), fp32 out.  All DEVICE pointers.
int dbm_op_conv2d_cl16(dbm_ctx* ctx, const float* x, const float* w, const float* b, const float* r1, float s1, float* y, int N,
                       int C, int H, int W, int O, int lrelu) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(C % 32 == 0 && C >= 32 && C <= 256 && (O == 32 || O == 64), "cl16 conv op: C % 32 == 0, O 32 or 64");
  DBM_CHECK(r1 == nullptr || O == 64, "cl16 conv op: the residual input has 64 channels");
  hipStream_t s = ctx->stream;
  const int plane = H * W;
  const size_t P = (size_t)N * plane;
  DevBuf act, res, out, wimg, bias;
  act.ensure(P * C / 2 + 8);
  out.ensure(P * 64);
  wimg.ensure(cl16_packed_elems(C, O) / 2 + 8);
  bias.ensure(64);
  if (b) DBM_HIP(hipMemcpyAsync(bias.p, b, sizeof(float) * O, hipMemcpyDeviceToDevice, s));
  for (int c0 = 0; c0 < C; c0 += 64)
    launch_nchw_to_cl(x + (size_t)c0 * plane, (long)C * plane, nullptr, (char*)act.p + 2 * (size_t)c0, C, N, plane, s, std::min(64, C - c0));
  if (r1) {
    res.ensure(P * 64);
    launch_nchw_to_cl(r1, 64L * plane, res.p, nullptr, 0, N, plane, s);
  }
  launch_pack_cl16(w, wimg.p, O, C, s);
  ClConvLaunch q;
  memset(&q, 0, sizeof(q));
  q.x = act.p; q.xc = C; q.Cin = C; q.Cout = O; q.w = wimg.p; q.bias = bias.p; q.y32 = out.p;
  q.r1 = r1 ? res.p : nullptr; q.s1 = r1 ? s1 : 1.f; q.s2 = 1.f; q.act = lrelu; q.slope = 0.2f; q.N = N; q.H = H; q.W = W;
  q.zeros = ctx->zeros;
  launch_conv_cl16(q, s);
  launch_cl_to_nchw(out.p, y, (long)O * plane, N, plane, s, O);
  DBM_HIP(hipStreamSynchronize(s));
  act.release(); res.release(); out.release(); wimg.release(); bias.release();
  DBM_API_END
}

// The split-bf16 form (conv_cl16x3_kernel: three bf16 MFMAs per product) of the same convolution, optionally on the nearest x2
// resize of x: x (N, 64, H >> ups, W >> ups), y (N, O, H, W), O <= 64; planar != 0 writes y through the kernel's channel-plane
// epilogue (the offset tensors' form) instead of NHWC + transpose.  All DEVICE pointers.
int dbm_op_conv2d_cl16x3(dbm_ctx* ctx, const float* x, const float* w, const float* b, float* y, int N, int H, int W, int O, int ups,
                         int lrelu, int planar) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(O >= 1 && O <= 64 && (ups == 0 || ups == 1), "cl16x3 conv op: O <= 64, ups 0 / 1");
  hipStream_t s = ctx->stream;
  const int Hs = H >> ups, Ws = W >> ups, plane = H * W, splane = Hs * Ws;
  DevBuf xin, out, wimg, bias;
  xin.ensure((size_t)N * splane * 64);
  out.ensure((size_t)N * plane * 64);
  wimg.ensure(cl16x3_packed_elems(64, O) / 2 + 8);
  bias.ensure(64);
  if (b) DBM_HIP(hipMemcpyAsync(bias.p, b, sizeof(float) * O, hipMemcpyDeviceToDevice, s));
  launch_nchw_to_cl(x, 64L * splane, xin.p, nullptr, 0, N, splane, s);
  launch_pack_cl16x3(w, wimg.p, O, 64, s);
  ClX3Launch q;
  memset(&q, 0, sizeof(q));
  q.x = xin.p; q.xc = 64; q.Cin = 64; q.Cout = O; q.ups = ups; q.w = wimg.p; q.bias = bias.p; q.act = lrelu; q.slope = 0.2f;
  q.N = N; q.H = H; q.W = W;
  if (planar) {
    DBM_HIP(hipMemsetAsync(y, 0, sizeof(float) * (size_t)N * O * plane, s));
    q.yp = y; q.ysn = (long)O * plane; q.ypc = O;
  } else {
    q.y32 = out.p; q.yc = 64;
  }
  launch_conv_cl16x3(q, s);
  if (!planar) launch_cl_to_nchw(out.p, y, (long)O * plane, N, plane, s, O);
  DBM_HIP(hipStreamSynchronize(s));
  xin.release(); out.release(); wimg.release(); bias.release();
  DBM_API_END
}

int dbm_op_deform_conv2d_form(dbm_ctx* ctx, const float* x, const float* off, const float* w, const float* b, float* y, int N, int H,
                              int W, int O, int form, int lrelu) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK((form == 1 && O >= 1 && O <= 16) || ((form == 2 || form == 3 || form == 4) && O == 64),
            "deform conv op: form 1 (O <= 16, premultiplied) or 2 (O = 64, split-bf16; 3: its LDS-window kernel, 4: its gathering kernel)");
  hipStream_t s = ctx->stream;
  const long plane = (long)H * W;
  DevBuf xt, z, wx;
  xt.ensure((size_t)N * 64 * plane);
  launch_nchw_to_nhwc64(x, xt.p, N, (int)plane, s);
  if (form == 1) {
    z.ensure((size_t)N * 9 * O * plane);
    launch_deform_conv_fused(xt.p, off, w, b, y, nullptr, nullptr, N, 64, H, W, 18L * plane, O, 0, 0.2f, s, z.p);
  } else {
    wx.ensure((deform_x3_packed_elems() + 1) / 2);
    launch_pack_deform_x3(w, wx.p, s);
    launch_deform_conv64_x3(xt.p, off, wx.p, b, y, nullptr, N, H, W, 18L * plane, lrelu, 0.2f, s, form == 3 ? 1 : form == 4 ? 0 : -1);
  }
  DBM_HIP(hipStreamSynchronize(s));
  xt.release(); z.release(); wx.release();
  DBM_API_END
}

int dbm_op_deform_conv2d(dbm_ctx* ctx, const float* x, const float* off, const float* w, const float* b, float* y,
                         int N, int C, int H, int W, int O) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(C % 32 == 0, "deform conv op: C % 32 == 0");
  static const int fused_env = DBM_TUNE_GETENV("DEFORM_FUSED") ? atoi(DBM_TUNE_GETENV("DEFORM_FUSED")) : 1;
  const bool fused = fused_env && deform_conv_fused_ok(C, O);
  DevBuf col;
  if (!fused) {
    col.ensure((size_t)N * C * 9 * H * W);
    launch_deform_sample(x, off, col.p, N, C, H, W, 18L * H * W, ctx->stream);
  }
  DevBuf xt;
  if (fused) {
    xt.ensure((size_t)N * C * H * W);
    launch_nchw_to_nhwc64(x, xt.p, N, H * W, ctx->stream);
  }
  if (fused && O <= 16) {  // (the few-output-channel form reads the OIHW weights directly, one launch per output channel)
    launch_deform_conv_fused(xt.p, off, w, b, y, nullptr, nullptr, N, C, H, W, 18L * H * W, O, 0, 0.2f, ctx->stream);
  } else if (O == 1) {
    launch_gemv_cols(col.p, w, b, y, N, C * 9, H * W, ctx->stream);
  } else {
    dbm_model holder;
    holder.ctx = ctx;
    holder.add_tensor("op/W", {O, C, 3, 3}, DBM_KIND_PARAM);
    holder.add_tensor("op/b", {O}, DBM_KIND_PARAM);
    holder.alloc_arenas();
    DBM_HIP(hipMemcpyAsync(holder.params, w, sizeof(float) * (size_t)O * C * 9, hipMemcpyDeviceToDevice, ctx->stream));
    if (b) DBM_HIP(hipMemcpyAsync(holder.params + (size_t)O * C * 9, b, sizeof(float) * O, hipMemcpyDeviceToDevice, ctx->stream));
    holder.add_iglayer("op", O, C, 3, 1, 0, true, true);
    holder.ensure_packed();
    if (fused) {
      launch_deform_conv_fused(xt.p, off, holder.layers[0].wf, b ? holder.P(holder.layers[0].bi) : nullptr, y, nullptr, nullptr, N, C, H, W,
                               18L * H * W, O, 0, 0.2f, ctx->stream);
    } else {
      ConvDesc d = holder.fwd_desc(holder.layers[0], col.p, (long)C * 9 * H * W, H, W, 0, y, (long)O * H * W, N);
      launch_igemm_conv(d, ctx->stream);
    }
    DBM_HIP(hipStreamSynchronize(ctx->stream));
  }
  DBM_HIP(hipStreamSynchronize(ctx->stream));
  col.release();
  xt.release();
  DBM_API_END
}

int dbm_op_deform_conv2d_backward(dbm_ctx* ctx, const float* x, const float* off, const float* w, const float* gy,
                                  float* gx, float* goff, float* gw, float* gb, int N, int C, int H, int W, int O) {
  DBM_API_BEGIN(ctx)
  DBM_CHECK(C % 32 == 0, "deform conv op: C % 32 == 0");
  hipStream_t s = ctx->stream;
  const long P = (long)H * W;
  static const int fused_env = DBM_TUNE_GETENV("DEFORM_FUSED") ? atoi(DBM_TUNE_GETENV("DEFORM_FUSED")) : 1;
  const bool fused = fused_env && deform_conv_fused_ok(C, O) && deform_input_grad_ok(C, H, W);
  DevBuf col, gcol, xt, part, cws;
  if (fused) {
    cws.ensure(deform_csr_workspace_floats(N, H, W));
    xt.ensure((size_t)N * C * P);
    launch_nchw_to_nhwc64(x, xt.p, N, (int)P, s);
  }
  if (!(fused && O == 1)) {
    col.ensure((size_t)N * C * 9 * P);
    launch_deform_sample(x, off, col.p, N, C, H, W, 18 * P, s);
  }
  if (O == 1) {
    if (fused) {
      part.ensure(deform_bwd1_partial_floats(N, H, W));
      // (the premultiplied form the generator's backward pass uses -- round 5 --; DBM_DEFORM1_PREMUL_BWD=0: the gathering kernels)
      static const bool premul_bwd = !(getenv("DBM_DEFORM1_PREMUL_BWD") && atoi(getenv("DBM_DEFORM1_PREMUL_BWD")) == 0);
      if (premul_bwd && C == 64) {
        DevBuf z, gt;
        z.ensure((size_t)N * 9 * P);
        gt.ensure((size_t)N * 9 * P);
        launch_deform1_premul(xt.p, w, z.p, N, H, W, 1, s);
        launch_deform_bwd1_premul(xt.p, off, w, gy, z.p, goff, gx, gw, gb, part.p, cws.p, gt.p, N, H, W, 18 * P, s);
        DBM_HIP(hipStreamSynchronize(s));
        z.release();
        gt.release();
      } else {
        launch_deform_bwd1_fused(xt.p, off, w, gy, goff, gw, gb, part.p, N, H, W, 18 * P, s);
        launch_deform_input_grad(x, off, nullptr, w, gy, gx, N, C, H, W, 18 * P, s, cws.p);
      }
    } else {
      launch_deform_backward(x, off, nullptr, w, gy, gx, goff, N, C, H, W, 18 * P, s);
      launch_gemv_cols_wgrad(col.p, gy, gw, gb, N, C * 9, (int)P, s);
    }
  } else {
    DBM_CHECK(O % 32 == 0, "deform conv op backward: O == 1 or O % 32 == 0");
    dbm_model holder;
    holder.ctx = ctx;
    holder.add_tensor("op/W", {O, C, 3, 3}, DBM_KIND_PARAM);
    holder.add_tensor("op/b", {O}, DBM_KIND_PARAM);
    holder.alloc_arenas();
    DBM_HIP(hipMemcpyAsync(holder.params, w, sizeof(float) * (size_t)O * C * 9, hipMemcpyDeviceToDevice, s));
    holder.add_iglayer("op", O, C, 3, 1, 0, true, true);
    holder.ensure_packed();
    holder.ensure_packed_bwd();
    const IgLayer& L = holder.layers[0];
    gcol.ensure((size_t)N * C * 9 * P);
    if (fused) {
      launch_deform_bwd64_fused(xt.p, off, L.wb[1], gy, gcol.p, goff, N, H, W, 18 * P, s);
      launch_deform_input_grad(x, off, gcol.p, nullptr, nullptr, gx, N, C, H, W, 18 * P, s, cws.p);
    } else {
      ConvDesc d;
      memset(&d, 0, sizeof(d));
      d.x = gy; d.xsn = O * P; d.N = N; d.y = gcol.p; d.ysn = C * 9 * P; d.s1 = 1.f; d.s2 = 1.f;
      holder.run_dgrad(L, d, H, W);
      launch_deform_backward(x, off, gcol.p, nullptr, nullptr, gx, goff, N, C, H, W, 18 * P, s);
    }
    // (the generator's own form since round 6: the sampler-fused weight gradient, no sample matrix; DBM_DEFORM_WGRAD_FUSED=0: the 1x1 form)
    static const int wg_fused_env = getenv("DBM_DEFORM_WGRAD_FUSED") ? atoi(getenv("DBM_DEFORM_WGRAD_FUSED")) : 1;
    if (fused && wg_fused_env && C == 64 && O == 64) {
      part.ensure(deform_wgrad64_partial_floats(N, H, W));
      launch_deform_wgrad64_fused(xt.p, off, gy, gw, gb, part.p, N, H, W, 18 * P, s);
    } else {
      WgradDesc wd;
      memset(&wd, 0, sizeof(wd));
      wd.x = col.p; wd.xsn = C * 9 * P; wd.xsc = (int)P; wd.Cin = C * 9; wd.Hin = H; wd.Win = W;
      wd.dy = gy; wd.dysn = O * P; wd.dysc = (int)P; wd.Cout = O; wd.OH = H; wd.OW = W;
      wd.KH = wd.KW = 1; wd.stride = 1; wd.pad = 0; wd.N = N; wd.scale = 1.f; wd.gW = gw; wd.gb = gb;
      launch_wgrad(wd, s);
    }
    DBM_HIP(hipStreamSynchronize(s));
  }
  DBM_HIP(hipStreamSynchronize(s));
  col.release();
  gcol.release();
  xt.release();
  part.release();
  cws.release();
  DBM_API_END
}

}  // extern "C"
