/* libdbm.so -- C ABI of the MI355X-native (gfx950) ESRGAN hot path of weiji14/deepbedmap.
 *
 * The reference has no FFI of its own: the path sits behind Python classes/functions of
 * srgan_train.py that call Chainer.  Each entry point below names the reference interface
 * (file:line under the reference checkout) it stands in for; INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add.  Conventions:
 *   - every function returns 0 on success, non-zero on error; dbm_last_error() gives the text;
 *     nothing throws across the boundary.  Status 7 is recoverable: a persistent kernel gave up waiting for a
 *     neighbouring workgroup (the GPU is shared or partitioned).  From that moment a sticky device flag turns every
 *     optimizer launch and every BatchNorm running-average write into a no-op, so no parameter, moment or running
 *     statistic absorbs the invalid pass.  The condition is observed ONLY at the entry of the step entry points
 *     (dbm_train_iteration, dbm_discriminator_step, dbm_generator_step, dbm_adam_update) and by dbm_check_timeout: the call
 *     drains the device, gives the skipped launches' step counts back, switches to the layer-by-layer trunk kernels for a
 *     while (re-armed after DBM_TRUNK_REARM = 64 iterations, doubling) and returns 7 WITHOUT having enqueued anything --
 *     re-issue it; dbm_timeout_info says how many queued updates were dropped.  Entry points that take HOST pointers and
 *     therefore end with a stream synchronisation (dbm_gen_forward / dbm_gen_backward -- the calls that launch persistent kernels --
 *     dbm_disc_forward and the loss calls, without DBM_DEVICE_PTRS) observe the condition after that synchronisation as well: their
 *     results are void, status 7.  FORWARD and loss calls are simply re-issued.  dbm_gen_backward / dbm_disc_backward are NOT: gradients
 *     accumulate, so after status 7 from a backward call: dbm_model_cleargrads, then repeat forward AND backward.
 *     dbm_adam_update is the other exception: whenever an event has been handled (by any call) since the model's gradient arena
 *     was last cleared, the arena may hold the sums of a void pass -- it returns status 9, applies nothing, and the caller clears
 *     the gradients and repeats forward + backward before updating (the step entry points clear them themselves).  Status 8:
 *     the same in a data-parallel run, where a local retry cannot keep the replicas identical -- fatal, abort the job;
 *   - tensors are NCHW float32, C-contiguous; weights OIHW, exactly the arrays stored by
 *     chainer.serializers.save_npz (key layout: SURVEY.md Appendix B);
 *   - pointers are HOST pointers unless flags contains DBM_DEVICE_PTRS, in which case they are
 *     device pointers on the context's GPU and the call only enqueues work on the context's
 *     stream (no synchronisation);
 *   - a dbm_ctx is bound to one GPU and one HIP stream and is not thread-safe; data parallelism
 *     is one process (one ctx) per GPU.
 */
#ifndef DBM_H
#define DBM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dbm_ctx dbm_ctx;
typedef struct dbm_model dbm_model;

enum {
  DBM_DEVICE_PTRS = 1, /* array arguments are device pointers; call is asynchronous on the ctx stream */
  DBM_KEEP_GRAPH = 2,  /* retain activations for a following backward (Chainer: enable_backprop=True) */
  DBM_BF16 = 8,        /* dbm_gen_forward without DBM_KEEP_GRAPH: the convolutions multiply in bf16 (operands rounded to
                          nearest-even, fp32 accumulation, fp32 storage): the area-inference mode of BASELINE config 5 */
  DBM_BN_TRAIN = 4,    /* discriminator BatchNorm uses batch statistics and updates running stats
                          (chainer.config.train=True, srgan_train.py:1125) */
  DBM_ONE_GEN_FORWARD = 16 /* dbm_train_iteration (opt-in, not the reference's call sequence): the generator runs ONCE per
                          minibatch -- the G-step's retained forward also supplies the D-step's fakes (same weights, same
                          inputs: srgan_train.py:1131-1137 and :1222-1227 compute the same images); bit for bit the two step calls
                          with their share flag, and the default iteration up to fp32 rounding (the unretained pass sums
                          conv_layer5 of the trunk in another order) */
};
enum { DBM_KIND_PARAM = 0, DBM_KIND_PERSISTENT = 1 };

/* ---- context ---- */
int dbm_init(int hip_device, dbm_ctx** out);      /* replaces model.to_gpu(): srgan_train.py:1038-1040, deepbedmap.py:659 */
int dbm_shutdown(dbm_ctx* ctx);
const char* dbm_last_error(dbm_ctx* ctx);         /* ctx may be NULL (error of a failed dbm_init) */
int dbm_set_stream(dbm_ctx* ctx, void* hip_stream); /* run on a caller-owned hipStream_t (NULL = the ctx's own) */
int dbm_synchronize(dbm_ctx* ctx);
/* chainer.global_config.cudnn_deterministic (srgan_train.py:69, deepbedmap.py:689).  on = 1: every gradient is folded in a
 * fixed order (no fp32 atomics over a K split: partial sums + an ordered fold kernel, sorted sampling lists, a separate
 * offset-gradient kernel), so a training run is bitwise reproducible; costs a few per cent.  Default 1, the reference's
 * setting.  Process-wide. */
int dbm_set_deterministic(dbm_ctx* ctx, int on);
/* sync_batch_stats (data-parallel training that must equal ONE process at the global batch): with world > 1 the
 * discriminator's training-mode BatchNorm layers (srgan_train.py:636-644, 663-689) use the statistics of the global batch in
 * forward and backward, and calculate_discriminator_loss (srgan_train.py:995-1004) the global-batch means of the logits.
 * The library computes per-rank sums into a small device buffer and calls allreduce_sum(user, dev, n), which must enqueue
 * an in-place SUM all-reduce of n floats on the context's stream (RCCL / torch.distributed on the shared stream).
 * With a native communicator of the same world (dbm_comm_init) the hook may be NULL: the sums go through RCCL.
 * world = 1 restores per-rank statistics, the default. */
int dbm_set_sync_batch_stats(dbm_ctx* ctx, int world, void (*allreduce_sum)(void* user, float* dev, int n), void* user);
/* ---- gradient exchange of a data-parallel run (one process / one dbm_ctx per GPU) ----
 * The reference trains on ONE GPU (srgan_train.py:58-61, 1039-1040: `model.to_gpu()` of a single device); these entry
 * points are what a multi-GPU `trainer` (srgan_train.py:1267-1329) calls between `backward()` and `optimizer.update()`
 * (:1163-1164, :1256-1257).  Native path: RCCL over xGMI, opened at run time (librccl.so.1), no torch involved.
 * dbm_comm_unique_id: rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other ranks by
 * any host channel; dbm_comm_init: every rank joins (collective call; world = 1 is allowed and makes everything below a
 * no-op).  With a communicator on the context, dbm_discriminator_step / dbm_generator_step sum their gradient arenas
 * over ranks THEMSELVES, bucket by bucket underneath the backward passes (D: conv_layer6..9 = 89 % of the bytes as soon
 * as their weight gradients are enqueued, the rest at the end; G: tail, trunk groups, input block), on a library
 * stream; the caller then runs dbm_adam_update(m, 1.0 / world).  Bit 4 (16) of `train` leaves the exchange to the caller.
 * dbm_comm_set_hook replaces RCCL by a callback (tests: several ranks on one GPU, where RCCL refuses to run): it must
 * enqueue an in-place SUM all-reduce of n floats ordered after the work already enqueued on hip_stream and before
 * work enqueued there later (a blocking implementation may synchronise that stream and reduce on the host). */
int dbm_comm_unique_id(void* out128);
int dbm_comm_init(dbm_ctx* ctx, int rank, int world, const void* id128);
int dbm_comm_set_hook(dbm_ctx* ctx, int rank, int world,
                      void (*allreduce_sum)(void* user, float* dev, size_t n, void* hip_stream), void* user);
int dbm_comm_destroy(dbm_ctx* ctx);
/* in-place broadcast / sum all-reduce of device floats on the context's stream (parameter broadcast at start-up; metrics) */
int dbm_comm_broadcast(dbm_ctx* ctx, float* dev, size_t nfloats, int root);
int dbm_comm_allreduce(dbm_ctx* ctx, float* dev, size_t nfloats);
/* bytes / collective calls issued so far on this context's communicator (reset != 0 clears the counters) */
int dbm_comm_stats(dbm_ctx* ctx, int* world, size_t* bytes, size_t* calls, int reset);

/* measurement aid (bench.py roofline leg): while enabled, every launch of the two MFMA kernel families is bracketed
 * by hipEvents on the launch stream.  out = [ms, algorithmic FLOP, launches] for igemm_conv_kernel (forward + data
 * gradient), then the same three for wgrad_kernel. */
int dbm_profile_begin(dbm_ctx* ctx);
int dbm_profile_end(dbm_ctx* ctx, double out[8]);
/* the same for nfam <= 5 kernel families, three values each: igemm_conv_kernel, the weight-gradient kernels,
 * trunk_fused_kernel (RRDB trunk forward, srgan_train.py:546), trunk_fused_bwd_kernel (its data-gradient chain),
 * trunk_fused_kernel in the form with a helper workgroup per image (passes that keep nothing; nfam <= 4: counted with the
 * third family) */
int dbm_profile_end_ex(dbm_ctx* ctx, double* out, int nfam);
/* the same brackets with the device synchronised before and after every bracketed launch: STANDALONE launch durations
 * (inside a training step up to four streams share the chip and every bracket also contains the neighbours' work).
 * Ended by dbm_profile_end_ex. */
int dbm_profile_begin_serial(dbm_ctx* ctx);
/* ends either kind of bracketing and returns EVERY bracket as a text line "family flops bytes ms wgs tag\n": family as in
 * dbm_profile_end_ex (0..4), the launch's algorithmic FLOP and algorithmic BYTES (operands read once + results written once),
 * its duration, its workgroup count (what joins a bracket to a rocprofv3 dispatch row) and a label of its shape (layer geometry) -- bench.py's per-shape roofline table, and the denominator of the
 * traffic ratio in profiles/<round>/traffic_pmc.json.  *len = the text's length; if it does not fit into cap (with its NUL)
 * nothing is copied and the text is kept for a second call with a larger buffer. */
int dbm_profile_end_records(dbm_ctx* ctx, char* buf, size_t cap, size_t* len);
/* testing aid: raises the condition a persistent trunk kernel raises when it gives up waiting for a neighbouring
 * workgroup.  From then on the optimizer launches and BatchNorm's running-average writes are no-ops; the next STEP entry
 * point (or dbm_check_timeout) returns status 7 without enqueuing anything. */
int dbm_debug_inject_timeout(dbm_ctx* ctx);
/* the same raised by a kernel ENQUEUED on the context's stream (no host synchronisation): the condition comes up in stream
 * order, as a persistent kernel's would, with whatever the host has queued behind it running under the raised flag */
int dbm_debug_inject_timeout_async(dbm_ctx* ctx);
/* Observes a pending persistent-kernel timeout like the step entry points do (status 7 / 8, see the conventions above); 0
 * when there is none.  For the end of an epoch: the last iterations of a run have no following step call that would
 * notice.  (srgan_train.py has no counterpart: Chainer's kernels cannot time out.) */
int dbm_check_timeout(dbm_ctx* ctx);
/* What the last handled timeout cost: number of events so far, optimizer updates of the discriminator / generator that
 * were no-ops (= minibatches whose update was dropped), and whether the persistent trunk kernels are currently paused.
 * Any pointer may be NULL. */
int dbm_timeout_info(dbm_ctx* ctx, long* events, int* d_updates_skipped, int* g_updates_skipped, int* persistent_off);
/* measurement aid: HIP-event stopwatch on the context's stream.  op 0 = record the start event, 1 = record the stop
 * event (both asynchronous), 2 = wait for the stop event and write the elapsed milliseconds to *ms. */
int dbm_timer(dbm_ctx* ctx, int op, double* ms);
/* measurement aid: while enabled, the step entry points record one hipEvent per phase boundary on the main stream;
 * enable = 0 stops, synchronises and writes "name milliseconds-since-the-first-mark" lines into out (cap bytes). */
int dbm_phase_marks(dbm_ctx* ctx, int enable, char* out, int cap);
int dbm_malloc(dbm_ctx* ctx, size_t bytes, void** dptr);
int dbm_free(dbm_ctx* ctx, void* dptr);
int dbm_memcpy_h2d(dbm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int dbm_memcpy_d2h(dbm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* device-resident area inference (deepbedmap.py:706-737: the per-tile `xp.asarray(...)` crops and the paste into Y_hat,
 * without leaving HBM): pitched device-to-device copy of `height` rows of `width_bytes`, and a float fill (NaN canvas).
 * Asynchronous on the context's stream. */
int dbm_memcpy2d_d2d(dbm_ctx* ctx, void* dst_dev, size_t dst_pitch, const void* src_dev, size_t src_pitch,
                     size_t width_bytes, size_t height);
int dbm_fill_f32(dbm_ctx* ctx, float* dst_dev, size_t n, float value);
/* `np.clip(a=W_tile, a_min=0.0, a_max=None)` (deepbedmap.py:663-665: ice surface elevation, velocity and accumulation are
 * clipped to >= 0 before the sweep) on a grid that already lives in HBM: in place, NaN stays NaN.  Asynchronous. */
int dbm_clip_min_f32(dbm_ctx* ctx, float* dst_dev, size_t n, float lo);
/* chainer.dataset.concat_examples over a dataset that lives on the device (srgan_train.py:107-121 `to_gpu`, 1286-1288):
 * dst row i = src row idx[i], rows of row_bytes (a multiple of 4) bytes; idx is a HOST array of n ints.  Asynchronous. */
int dbm_gather_rows(dbm_ctx* ctx, void* dst_dev, const void* src_dev, const int* idx_host, int n, size_t row_bytes);

/* ---- models ---- */
/* GeneratorModel.__init__(num_residual_blocks=12, residual_scaling=0.1, out_channels=1): srgan_train.py:450-523.
 * Parameters are created zero; the caller uploads them (HeNormal init or load_npz) with dbm_model_set_tensor.
 * out_channels in [1, 16]; more than one channel is forward-only (y is (N, out_channels, 4(H-2), 4(W-2))): the reference's
 * own training step fails with it (F.mean_absolute_error against the one-channel x_topo, srgan_train.py:882-883). */
int dbm_gen_create(dbm_ctx* ctx, int num_residual_blocks, float residual_scaling, int out_channels, dbm_model** out);
/* DiscriminatorModel.__init__(): srgan_train.py:611-647 */
int dbm_disc_create(dbm_ctx* ctx, dbm_model** out);
int dbm_model_destroy(dbm_model* m);
/* Link.namedparams()/serialize(): tensors in chainer.serializers.save_npz key order -- srgan_train.py:1355-1361, deepbedmap.py:408 */
int dbm_model_num_tensors(dbm_model* m, int* n);
int dbm_model_tensor_info(dbm_model* m, int i, const char** npz_key, int* ndim, int64_t shape[4], int* kind);
int dbm_model_set_tensor(dbm_model* m, const char* npz_key, const float* host, size_t nfloats);
int dbm_model_get_tensor(dbm_model* m, const char* npz_key, float* host, size_t nfloats);
int dbm_model_get_grad(dbm_model* m, const char* npz_key, float* host, size_t nfloats);
/* Link.count_params(): srgan_train.py:446, 607 */
int dbm_model_count_params(dbm_model* m, int64_t* n);
/* Link.cleargrads(): srgan_train.py:1162, 1255 */
int dbm_model_cleargrads(dbm_model* m);
/* flat fp32 arenas (device pointers) holding every parameter / gradient contiguously in tensor order:
 * what a data-parallel host all-reduces (RCCL) between backward and update */
int dbm_model_param_arena(dbm_model* m, void** dptr, size_t* nfloats);
int dbm_model_grad_arena(dbm_model* m, void** dptr, size_t* nfloats);
/* tell the library the parameter arena was written from outside (e.g. an RCCL broadcast) so that its packed
 * MFMA weight images are rebuilt before the next forward */
int dbm_model_params_changed(dbm_model* m);

/* ---- forward / backward ---- */
/* GeneratorModel.forward(x, w1, w2, w3): srgan_train.py:525-576.  x (N,1,H,W), w1 (N,1,10H,10W), w2 (N,2,2H,2W),
 * w3 (N,1,H,W) -> y (N,1,4(H-2),4(W-2)).  flags: DBM_DEVICE_PTRS, DBM_KEEP_GRAPH, DBM_BF16. */
int dbm_gen_forward(dbm_model* g, int N, int H, int W, const float* x, const float* w1, const float* w2,
                    const float* w3, float* y, int flags);
/* g_loss.backward() through the generator: srgan_train.py:1256.  gy (N,1,4(H-2),4(W-2)) = d loss / d y of the last
 * DBM_KEEP_GRAPH forward; accumulates into the gradient arena. */
int dbm_gen_backward(dbm_model* g, const float* gy, int flags);
/* DiscriminatorModel.forward(x): srgan_train.py:649-699.  img (N,1,36,36) -> logits (N,1).
 * slot (0/1) selects which retained graph a DBM_KEEP_GRAPH call fills (the D-step runs real and fake batches). */
int dbm_disc_forward(dbm_model* d, int N, int H, int W, const float* img, float* logits, int flags, int slot);
int dbm_disc_backward(dbm_model* d, int slot, const float* glogits, int flags);

/* ---- losses / metrics ---- */
/* calculate_discriminator_loss: srgan_train.py:960-1009 (+ F.binary_accuracy :1156-1158).
 * out[0] = loss, out[1] = accuracy; g_real/g_fake (N) may be NULL. */
int dbm_discriminator_loss(dbm_ctx* ctx, const float* real_logits, const float* fake_logits, int N,
                           int real_minus_fake_target, int fake_minus_real_target, float* out2, float* g_real,
                           float* g_fake, int flags);
/* the same with per-sample int32 target arrays (N each; 0, 1 or -1 = ignored), what the reference's signature accepts
 * (srgan_train.py:960-1004: `real_minus_fake_target`, `fake_minus_real_target` are arrays handed to
 * F.sigmoid_cross_entropy, each call normalised by its count of targets != -1) */
int dbm_discriminator_loss_t(dbm_ctx* ctx, const float* real_logits, const float* fake_logits, int N,
                             const int* real_minus_fake_target, const int* fake_minus_real_target, float* out2, float* g_real,
                             float* g_fake, int flags);
/* calculate_generator_loss: srgan_train.py:841-902, psnr :906-928, ssim_loss_func :932-956.
 * y_pred,y_true (N,1,H,W); x (N,1,H/4+2,W/4+2) is the full BEDMAP2 tile (x_topo = x[:,:,1:-1,1:-1], :1248);
 * fake_logits (N) from the discriminator in eval mode, real_logits (N) or NULL for the reference's ones(N) (:1233);
 * the adversarial term is calculate_discriminator_loss(real, fake, real_minus_fake_target, fake_minus_real_target)
 * (:874-879; the G-step passes targets 0 and 1, :1236-1237).  weights[4] = content, adversarial, topographic,
 * structural.  out[0] = g_loss, out[1] = psnr, out[2] = ssim; gy (N,1,H,W) may be NULL (the adversarial term is
 * detached from y_pred, :1228-1229).  ssim_window: 0 gaussian(1.5), 1 uniform. */
int dbm_generator_loss(dbm_ctx* ctx, const float* y_pred, const float* y_true, const float* x,
                       const float* real_logits, const float* fake_logits, int N, int H, int W,
                       const float weights[4], int real_minus_fake_target, int fake_minus_real_target,
                       int ssim_window, float* out3, float* gy, int flags);
/* the same with per-sample int32 target arrays for the adversarial term (see dbm_discriminator_loss_t) */
int dbm_generator_loss_t(dbm_ctx* ctx, const float* y_pred, const float* y_true, const float* x,
                         const float* real_logits, const float* fake_logits, int N, int H, int W,
                         const float weights[4], const int* real_minus_fake_target, const int* fake_minus_real_target,
                         int ssim_window, float* out3, float* gy, int flags);

/* psnr(y_pred, y_true, data_range=2**32): srgan_train.py:906-928 over n elements; out[0] = 20*log10(range/sqrt(mse)) */
int dbm_psnr(dbm_ctx* ctx, const float* y_pred, const float* y_true, size_t n, double data_range, float* out,
             int flags);
/* ssim_loss_func(y_pred, y_true, window_size=9, stride=1): srgan_train.py:932-956; (N,1,H,W), H,W >= 9 */
int dbm_ssim(dbm_ctx* ctx, const float* y_pred, const float* y_true, int N, int H, int W, int ssim_window,
             float* out, int flags);
/* ssim_loss_func with any window_size (1..64, gaussian(1.5) centred at window_size / 2 or uniform) and stride >= 1:
 * the metric for other windows than the loss's 9 / 1 (valid windows only; N counts images x channels) */
int dbm_ssim_ex(dbm_ctx* ctx, const float* y_pred, const float* y_true, int N, int H, int W, int window_size, int stride,
                int ssim_window, float* out, int flags);

/* ---- optimizer ---- */
/* chainer.optimizers.Adam(alpha, eps=1e-8).setup(model): srgan_train.py:1043-1048 */
int dbm_adam_setup(dbm_model* m, double alpha, double beta1, double beta2, double eps);
/* optimizer.update(): srgan_train.py:1164, 1257.  grad_scale multiplies the gradient first (1/world after a
 * sum all-reduce). */
int dbm_adam_update(dbm_model* m, double grad_scale);

/* Whole-arena gradient all-reduce on the context's stream for callers that drive dbm_gen_backward / dbm_disc_backward
 * themselves (no overlap); *grad_scale (may be NULL) receives 1 / world for dbm_adam_update. */
int dbm_allreduce_grads(dbm_model* m, double* grad_scale);

/* ---- fused steps (device-resident inputs, asynchronous) ---- */
/* train_eval_discriminator: srgan_train.py:1084-1166 up to and including d_loss.backward() (update = dbm_adam_update).
 * arrays are DEVICE pointers: X (N,1,11,11), W1 (N,1,110,110), W2 (N,2,22,22), W3 (N,1,11,11), Y (N,1,36,36).
 * metrics_dev (device, >= 8 floats) receives [d_loss, d_accu].
 * train: bit 0 = training mode (0 evaluates with BatchNorm in eval mode); scheduling options, all numerically neutral:
 * bit 1 (2) = keep this call's generator forward for the following dbm_generator_step on the same arrays (that step
 * then skips its own forward: NOT what the reference does, off by default); bit 2 (4) = the following
 * dbm_generator_step's forward is enqueued now, in its own workspace and on separate streams, underneath this
 * step's discriminator passes (the trainer's pattern; discarded if the next call does not match); bit 3 (8) = the
 * caller runs collectives on a stream of its own: the prefetched forward stays on one library stream; bit 4 (16) = do
 * not exchange gradients inside the call although the context has a communicator (dbm_comm_init). */
int dbm_discriminator_step(dbm_model* g, dbm_model* d, int N, int H, int W, const float* X, const float* W1,
                           const float* W2, const float* W3, const float* Y, int train, float* metrics_dev);
/* train_eval_generator: srgan_train.py:1170-1263 up to and including g_loss.backward().
 * metrics_dev receives [., ., g_loss, psnr, ssim].
 * train: bit 0 = training mode; bit 1 (2) = reuse the generator forward the preceding dbm_discriminator_step kept
 * (its bit 1); bit 2 (4) = consume the forward that step prefetched (its bit 2): the caller asserts that the five
 * arrays are the same, UNCHANGED, device arrays.  The library additionally checks pointers, shapes, the parameter
 * version and its own record of writes to device memory (dbm_memcpy_h2d, dbm_gather_rows, dbm_fill_f32,
 * dbm_memcpy2d_d2d, dbm_malloc, dbm_free); writes by anybody else (another library filling the same buffer in
 * place) are invisible to it, hence the explicit bit.  Without it the prefetched pass is discarded and the forward is
 * recomputed.  bit 4 (16) = see dbm_discriminator_step. */
int dbm_generator_step(dbm_model* g, dbm_model* d, int N, int H, int W, const float* X, const float* W1,
                       const float* W2, const float* W3, const float* Y, const float weights[4], int ssim_window,
                       int train, float* metrics_dev);

/* ---- output format of the DEM (deepbedmap.py:749-756: `save_array_to_grid(array=Y_hat.astype(np.int16), dtype=np.int16,
 * tiled=True, compression=lzw)` -> data_prep.py:779-834, a tiled LZW GeoTIFF written by rasterio / GDAL) ----
 * dbm_f32_to_i16: `Y_hat.astype(np.int16)` on the device canvas (NumPy's cast: truncation, NaN / inf / out of range -> 0),
 * asynchronous on the context's stream; dst_dev holds n int16 values.
 * dbm_lzw_encode_tiles: TIFF 6.0 LZW of `ntiles` tiles of `tile_bytes` bytes each (host memory, tile t at
 * tiles + t * tile_bytes) into out + t * out_stride (out_stride >= tile_bytes * 3 / 2 + 64 is always enough),
 * encoded sizes in out_sizes[t]; tiles are spread over `nthreads` host threads.  dbm_lzw_decode: one stream back
 * (round-trip check).  Host functions: no GPU, no context.  The TIFF container is written by the host shim
 * (deepbedmap_amd/geotiff.py). */
int dbm_f32_to_i16(dbm_ctx* ctx, const float* src_dev, void* dst_dev, size_t n);
int dbm_lzw_encode_tiles(const void* tiles, size_t tile_bytes, int ntiles, void* out, size_t out_stride, size_t* out_sizes,
                         int nthreads);
int dbm_lzw_decode(const void* src, size_t nbytes, void* dst, size_t cap, size_t* out_bytes);

/* One minibatch of `trainer` (srgan_train.py:1286-1309) as ONE call: train_eval_discriminator (:1084-1166) with its
 * optimizer update, then train_eval_generator (:1170-1263) with its update; both optimizers must have been set up
 * (dbm_adam_setup).  Numerically the two step calls + two dbm_adam_update calls, bit for bit; scheduled as a whole: the
 * generator's backward pass -- independent of everything the D-step computes, since the adversarial term is detached
 * (:1228-1229) -- runs on a library stream underneath the discriminator's backward passes.  metrics_dev (device, >= 8
 * floats) receives [d_loss, d_accu, g_loss, psnr, ssim].  With a communicator on the context (dbm_comm_init /
 * dbm_comm_set_hook) the call is one data-parallel iteration: both models' gradient buckets are summed over ranks inside
 * it (on library stream chain[0], underneath the generator's backward pass) and both updates take 1 / world -- the same
 * collectives in the same order as the two step calls.  Refused with sync_batch_stats (use the two step calls).
 * flags: 0 or DBM_ONE_GEN_FORWARD.
 * DEFERRED METRIC (round 6; opt-in: DBM_ITER_DEFER_EVAL=1 in the environment -- measured slower inside a training loop, see api.hip):
 * g_loss / psnr / ssim (metrics_dev[2..4]) need the G-step's detached eval-mode discriminator pass
 * (:1228-1237), which feeds that logged value and nothing else.  The call then snapshots what the pass reads (eval-mode BatchNorm
 * coefficients behind the discriminator's update, the fakes, the loss terms' sums) and the NEXT library call on the context
 * enqueues the pass itself: the next dbm_train_iteration beside its generator forwards, or ANY other entry point (dbm_synchronize,
 * dbm_memcpy_d2h, ...) on the context's stream before doing its own work.  So: metrics_dev[0..1] are complete when this call's work
 * is; metrics_dev[2..4] once the next library call's work is -- read them through the library (dbm_memcpy_d2h, or dbm_synchronize
 * first), and keep metrics_dev allocated until then.  Bitwise the same numbers (same kernels, same inputs).  Default: the pass runs
 * inside the call and all five metrics are complete when its work is. */
int dbm_train_iteration(dbm_model* g, dbm_model* d, int N, int H, int W, const float* X, const float* W1, const float* W2,
                        const float* W3, const float* Y, const float weights[4], int ssim_window, int flags,
                        float* metrics_dev);

/* ---- op-level entry points (used by the parity tests; same kernels the models run) ---- */
/* L.Convolution2D forward on the MFMA implicit-GEMM kernel. x (N,C,H,W) w (O,C,k,k) b (O) or NULL -> y; all DEVICE. */
int dbm_op_conv2d(dbm_ctx* ctx, const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W,
                  int O, int k, int stride, int pad, int upsample2, int lrelu);
/* data gradient (gx, may be NULL) and weight/bias gradient (gw, gb accumulated; may be NULL) of the same layer */
int dbm_op_conv2d_backward(dbm_ctx* ctx, const float* x, const float* w, const float* gy, float* gx, float* gw,
                           float* gb, int N, int C, int H, int W, int O, int k, int stride, int pad, int upsample2);
/* the same L.Convolution2D (3x3, stride 1, pad 1: the RRDB trunk's layers, srgan_train.py:292-331) on the channels-last
 * bf16 kernel of the area sweep (conv_cl16.hip): x is rounded to bf16, fp32 accumulation, y = [lrelu](s1 * (conv + b) + r1)
 * with r1 (N,64,H,W) or NULL; C % 32 == 0, O = 32 or 64 */
int dbm_op_conv2d_cl16(dbm_ctx* ctx, const float* x, const float* w, const float* b, const float* r1, float s1, float* y, int N,
                       int C, int H, int W, int O, int lrelu);
/* ... and in the sweep's split-bf16 arithmetic (three bf16 MFMAs per product: operands carry 16 significand bits) for the
 * layers on the signal path -- post_upsample_conv_layer_1/2 behind F.resize_images (srgan_train.py:553-568; ups = 1: x is the
 * (H/2, W/2) plane) and the deformable layers' offset convolutions (:506-523; planar = 1: channel-plane output):
 * x (N,64,H>>ups,W>>ups) -> y (N,O,H,W), O <= 64 */
int dbm_op_conv2d_cl16x3(dbm_ctx* ctx, const float* x, const float* w, const float* b, float* y, int N, int H, int W, int O, int ups,
                         int lrelu, int planar);
/* L.DeformableConvolution2D sampler + GEMM (stride 1, pad 1, 3x3): off (N,18,H,W) */
int dbm_op_deform_conv2d(dbm_ctx* ctx, const float* x, const float* off, const float* w, const float* b, float* y,
                         int N, int C, int H, int W, int O);
/* the two other forms of the forward pass the generator uses, 64 input channels: form 1 = the few-output-channel layer
 * (O <= 16; srgan_train.py:574, the DEM itself) with the multiplication BEFORE the sampler -- nine premultiplied tap planes,
 * scalar gathers --, form 2 = the 64 -> 64 layer (:572) in the sweep's split-bf16 arithmetic (+ LeakyReLU 0.2 if lrelu); forms 3 and 4
 * name form 2's two kernels explicitly -- 3: the sampler reads an LDS window of the input (what the sweep's full-resolution planes take),
 * 4: it gathers every corner from memory (small planes); same arithmetic, same bits */
int dbm_op_deform_conv2d_form(dbm_ctx* ctx, const float* x, const float* off, const float* w, const float* b, float* y, int N, int H,
                              int W, int O, int form, int lrelu);
int dbm_op_deform_conv2d_backward(dbm_ctx* ctx, const float* x, const float* off, const float* w, const float* gy,
                                  float* gx, float* goff, float* gw, float* gb, int N, int C, int H, int W, int O);

#ifdef __cplusplus
}
#endif
#endif
