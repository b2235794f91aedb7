// Model objects behind the C ABI: parameter tables in chainer save_npz key order, packed-weight
// caches for the MFMA kernels, activation workspaces, forward/backward orchestration.
#pragma once
#include "dbm_internal.h"
#include "kernels.h"
#include "../../include/dbm.h"

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
  void ensure(size_t count, bool zero = false);
  void release();
};

struct dbm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  hipStream_t side = nullptr;        // library-owned side stream: independent work overlapping the main chain
  hipEvent_t ev_fork[8] = {};        // main -> side dependencies
  hipEvent_t ev_join = nullptr;      // side -> main
  hipStream_t chain[2] = {};         // library-owned streams: second image range of the 9x9 stage / fake-batch D backward;
                                     // main stream of a prefetched generator forward (never more than four streams busy)
  void fork(hipStream_t from, hipStream_t to, int k);  // `to` waits for everything enqueued on `from` so far
  void fork_to_side(int k);          // side waits for everything enqueued on `stream` so far
  void join_side();                  // `stream` waits for everything enqueued on `side` so far
  std::string err;
  // sync_batch_stats (dbm_set_sync_batch_stats): > 1 = BatchNorm / RaGAN statistics are summed over `sync_world` ranks by
  // the caller's hook, which must enqueue the collective on `stream`
  int sync_world = 1;
  void (*sync_fn)(void* user, float* dev, int n) = nullptr;
  void* sync_user = nullptr;
  bool sync_stats() const { return sync_world > 1 && (sync_fn != nullptr || nccl_comm != nullptr); }
  void allreduce(float* dev, int n);  // statistics collective on `stream`: the hook, or the native communicator
  DevBuf sync_buf;            // 3 x 512 floats of per-channel sums + 4 for the loss
  // ---- gradient exchange of a data-parallel run (comm.hip): native RCCL communicator, or a caller hook ----
  void* nccl_comm = nullptr;  // ncclComm_t (dbm_comm_init)
  int comm_rank = 0, comm_world = 1;
  void (*comm_hook)(void* user, float* dev, size_t n, void* hip_stream) = nullptr;  // dbm_comm_set_hook
  void* comm_user = nullptr;
  bool comm_in_step = false;  // set by the fused steps: the backward passes hand finished buckets to comm_bucket
  hipStream_t comm_stream = nullptr;  // stream the collectives are enqueued on (null: chain[1]; dbm_train_iteration: chain[0])
  bool comm_defer = false;    // comm_bucket only records the producers' event; the all-reduce goes out with comm_flush()
  struct PendingBucket { float* p[2]; size_t n[2]; int nranges; hipEvent_t ev; };
  std::vector<PendingBucket> comm_pending;
  std::vector<hipEvent_t> comm_ev_pool;  // one event per bucket of a pass (reused from pass to pass)
  size_t comm_ev_used = 0;
  void comm_flush();
  hipEvent_t ev_comm = nullptr, ev_comm_done = nullptr;
  hipEvent_t ev_timer[2] = {nullptr, nullptr};  // dbm_timer
  hipEvent_t ev_iter[4] = {nullptr, nullptr, nullptr, nullptr};   // dbm_train_iteration: loss scratch cleared / generator backward done /
                                                                // G grads cleared + its data-gradient images / D's data-gradient images
  // The persistent trunk kernels need every workgroup of a launch resident at once: two of them on different streams, each
  // holding part of the chip, would wait for each other's compute units until their spin limits.  Every persistent launch
  // therefore waits for the previous one (whatever its stream) and leaves its own completion here.
  hipEvent_t ev_persist = nullptr;
  bool persist_pending = false;
  void persist_begin(hipStream_t s);
  void persist_end(hipStream_t s);
  size_t comm_bytes = 0, comm_calls = 0;  // statistics (dbm_comm_stats)
  // DBM_COMM_FORCE_WORLD1=1 (testing / measuring the data-parallel schedule on ONE GPU): a one-rank communicator counts as
  // active -- every bucket really goes through ncclAllReduce (a sum over one rank), the persistent launches keep to 192
  // workgroups, the optimizers wait for the exchange events
  static bool comm_force_world1() {
    static const bool f = getenv("DBM_COMM_FORCE_WORLD1") && atoi(getenv("DBM_COMM_FORCE_WORLD1")) != 0;
    return f;
  }
  bool comm_active() const {
    return (comm_world > 1 || (comm_force_world1() && comm_world == 1 && nccl_comm != nullptr)) && (nccl_comm != nullptr || comm_hook != nullptr);
  }
  void comm_init(int rank, int world, const void* id128);
  void comm_set_hook(int rank, int world, void (*fn)(void*, float*, size_t, void*), void* user);
  void comm_destroy();
  void comm_allreduce(float* const* p, const size_t* n, int nranges, hipStream_t on);
  void comm_broadcast(float* p, size_t n, int root, hipStream_t on);
  void comm_bucket(float* const* p, const size_t* n, int nranges, hipStream_t producer);
  void comm_bucket(float* p, size_t n, hipStream_t producer) { comm_bucket(&p, &n, 1, producer); }
  void comm_join(hipStream_t consumer);
  int trunk_imgs = 64;        // images per launch of the persistent trunk kernels: min(64, CUs / 3) (one workgroup per CU)
  std::vector<struct dbm_model*> models;  // every model of this context (optimizer bookkeeping after a kernel timeout)
  long data_epoch = 0;        // bumped by every entry point that writes / frees caller-visible device memory
  int* dev_err = nullptr;     // host-mapped word a persistent kernel raises when a bounded spin runs out (checked by every API call)
  int* dev_err_d = nullptr;   // its device address
  int* dev_err_flag = nullptr;  // the same flag in device memory, STICKY until the host handles it (the optimizer launches and
                                // BatchNorm's running-average writes are no-ops while it is set)
  long timeout_events = 0;      // persistent-kernel timeouts handled so far (dbm_timeout_info)
  int timeout_skipped[2] = {0, 0};  // optimizer updates (discriminator, generator) that were no-ops in the last event
  float* zeros = nullptr;     // 256 B of zeros (igemm out-of-image taps)
  float* ssim_win[2] = {nullptr, nullptr};  // 9-tap 1-D windows: gaussian(1.5), uniform
  DevBuf loss_tmp;            // scratch for the loss entry points
  // dbm_train_iteration with DBM_ITER_DEFER_EVAL=1 (round 6, opt-in): the G-step's detached eval-mode discriminator pass (srgan_train.py:1228) feeds a LOGGED value only
  // (the adversarial term of g_loss; its gradient never reaches the generator).  Iteration i therefore only SNAPSHOTS what that pass
  // reads -- the eval-mode BatchNorm coefficients right behind the discriminator's update (parameters and running statistics as the
  // reference's call sees them), the generator's fakes, the loss terms' partial sums -- and the pass itself (nine convolutions, two
  // linear layers, the loss value -> that iteration's metrics row) is enqueued by the NEXT library call: inside iteration i + 1 beside
  // its generator forwards, or on the main stream at the entry of any other entry point (every DBM_API_BEGIN flushes; dbm_synchronize
  // and every copy the caller reads metrics with are entry points).  Same kernels, same inputs: the logged numbers are bitwise the same.
  struct DeferredEval {
    bool pending = false;
    struct Discriminator* d = nullptr;
    int N = 0, H4 = 0, W4 = 0;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    float* out3 = nullptr;          // device: [g_loss, psnr, ssim] of that iteration's metrics row
    DevBuf fakes, scratch, logits;  // scratch: loss_tmp's layout (16 + 5 N floats)
    hipEvent_t ev_ready = nullptr;  // the snapshots are complete (recorded on the main stream)
  } deferred;
  DevBuf stage[8];            // host<->device staging for the non-DEVICE_PTRS entry points
};

void dbm_comm_unique_id_impl(void* out128);  // comm.hip: ncclGetUniqueId
// A persistent trunk kernel timed out: the layer-by-layer trunk path serves until iteration g_trunk_rearm_at (-1: for good);
// g_step_serial counts the training iterations entered (api.hip: dbm_step_entry).
extern bool g_trunk_fused_off;
extern long g_step_serial, g_trunk_rearm_at;

struct Tensor {
  std::string key;
  int ndim;
  int64_t shape[4];
  int kind;
  size_t off;  // float offset inside the param (kind 0) or persistent (kind 1) arena
  size_t n;
};

// one L.Convolution2D executed by the implicit-GEMM kernel
struct IgLayer {
  int wi = -1, bi = -1;  // tensor indices (bias may be -1)
  int O = 0, C = 0, K = 0, stride = 1, pad = 1;
  int Cview = 0, Kview = 0;  // the (C, K) the GEMM sees (deform_conv: C*9, 1)
  int CinP = 0, CoutP = 0;
  float* wf = nullptr;       // [T][CinP][CoutP]
  void* wf16 = nullptr;      // bf16 [T][CinP/16][2][CoutP][8]: a lane's eight K values of one MFMA are one 16-byte load
  bool want_cl16 = false;    // also a fragment-ordered bf16 image for the channels-last kernel (conv_cl16.hip): trunk layers
  void* wcl16 = nullptr;     // bf16 [chunk][tap][k half][mtile][lane][8]
  bool want_x3 = false;      // ... and a split-bf16 (hi | lo) image for conv_cl16x3_kernel: the sweep's upsampling / offset convs
  void* wx3 = nullptr;       // bf16 [chunk][tap][mtile][hi | lo][lane][8]
  bool want_dx3 = false;     // the 64 -> 64 deformable layer: split-bf16 image for deform_conv64_x3_kernel (launch_pack_deform_x3)
  void* wdx3 = nullptr;
  int OP = 0, CP = 0;
  float* wb[4] = {nullptr, nullptr, nullptr, nullptr};  // dgrad packs [Tb][OP][CP]
  int Tb = 0;
  bool lazy = false;  // packed on demand only (dbm_model::ensure_packed_lazy)
  signed char bky[4][DBM_MAX_TAPS], bkx[4][DBM_MAX_TAPS], bdy[4][DBM_MAX_TAPS], bdx[4][DBM_MAX_TAPS];
};

struct dbm_model {
  dbm_ctx* ctx = nullptr;
  int type = 0;  // 0 generator, 1 discriminator
  std::vector<Tensor> tensors;
  std::map<std::string, int> index;
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  size_t nparam = 0;
  float* pers = nullptr;
  size_t npers = 0;
  double alpha = 1.6e-4, beta1 = 0.9, beta2 = 0.999, eps = 1e-8;
  long adam_t = 0;
  int* d_adam_skipped = nullptr;  // device counter of optimizer launches that were no-ops (dbm_ctx::dev_err_flag was set)
  bool adam_ready = false;
  // a persistent-kernel time-out was handled while this model's gradient arena held (or may have held) the sums of a void
  // backward pass: dbm_adam_update answers status 9 until the arena has been cleared (dbm_model_cleargrads, or the cleargrads
  // inside the step entry points) -- whichever call observed the event, and however many host-synchronising calls lie between
  bool grads_void = false;
  // a backward pass has been enqueued into this model's gradient arena since it was last cleared (set by Generator / Discriminator::backward
  // for every model sharing the arena, reset by mark_grads_cleared): a time-out observed while the arena is still clean voids nothing
  bool grads_touched = false;
  void mark_grads_touched();
  bool packed_dirty = true;
  long param_version = 0;  // bumped by every write to the parameter arena
  long packed16_version = -1;  // param_version the bf16 forward images were built from
  bool use_bf16 = false;       // fwd_desc hands out the bf16 images (set around a DBM_BF16 forward)
  void ensure_packed_bf16();
  bool is_view = false;    // arenas and packed weight images belong to another model (Generator::twin)
  std::vector<IgLayer> layers;
  PackJob* d_pack_jobs = nullptr;  // device job table of the one-launch weight repack: the FORWARD images (IgLayer::wf)
  int n_pack_jobs = 0, n_pack_blocks = 0;
  // The data-gradient images (IgLayer::wb) are first read by the NEXT backward pass, not by the forward pass that follows an update --
  // in dbm_train_iteration the discriminator's eval-mode pass right behind its update: they have their own table and are rebuilt by
  // ensure_packed_bwd (head of the next iteration, side stream; every backward entry point calls it as well).  DBM_PACK_SPLIT=0: one
  // launch builds both, as before round 5.
  PackJob* d_bwd_jobs = nullptr;
  int n_bwd_jobs = 0, n_bwd_blocks = 0;
  bool bwd_dirty = true;
  void ensure_packed_bwd(hipStream_t on = nullptr);
  PackJob* d_lazy_jobs = nullptr;  // the same for the layers marked lazy
  int n_lazy_jobs = 0, n_lazy_blocks = 0;
  bool pack_tables_built = false;
  long lazy_version = -1;          // param_version the lazy layers' images were built from
  virtual ~dbm_model();
  int add_tensor(const std::string& key, std::vector<int64_t> shape, int kind);
  void alloc_arenas();
  float* P(int ti) const { return params + tensors[ti].off; }
  float* G(int ti) const { return grads + tensors[ti].off; }
  float* S(int ti) const { return pers + tensors[ti].off; }
  int tid(const std::string& key) const;
  int add_iglayer(const std::string& name, int O, int C, int K, int stride, int pad, bool bias, bool as_1x1 = false);
  void ensure_packed(hipStream_t on = nullptr);  // rebuild the packed weight images if the parameters changed
  void ensure_packed_lazy(hipStream_t on = nullptr);  // ... including the layers marked lazy
  virtual void pack_extra(hipStream_t) {}        // model-specific images, same launch point
  // helpers building descriptors
  ConvDesc fwd_desc(const IgLayer& L, const float* x, long xsn, int Hin, int Win, int ups, float* y, long ysn, int N) const;
  void run_dgrad(const IgLayer& L, ConvDesc base, int Hin_fwd, int Win_fwd, hipStream_t s = nullptr) const;
  // weight gradient of layer L: queued on `batch` (launched later, all layers at once) or run immediately
  void run_wgrad(const IgLayer& L, const float* x, long xsn, int Hin, int Win, int ups, const float* dy, long dysn,
                 int OH, int OW, int N, float scale, WgradBatch* batch = nullptr) const;
};

struct Generator : dbm_model {
  int n_rrdb = 12, out_ch = 1;
  float rs = 0.1f;
  // layer indices into `layers`
  int L_pre, L_post, L_up1, L_up2, L_off1, L_def1, L_off2;
  std::vector<int> L_rdb;  // nrdb*5
  int T_in[4][2];          // input block (W, b) tensor ids
  int L_in[4] = {-1, -1, -1, -1};  // W1 / W2 branches run as im2col + GEMM on the MFMA kernels
  DevBuf colW1, colW2;
  int T_def2W, T_def2b;
  // workspace
  int wsN = 0, wsH = 0, wsW = 0;
  bool wsTrain = false;
  bool have_graph = false;
  // what the retained graph was computed from (opt-in reuse of the D-step's generator forward by the G-step)
  long graph_version = -1;
  long graph_epoch = -1;   // dbm_ctx::data_epoch at the time of that forward
  const float* graph_in[4] = {nullptr, nullptr, nullptr, nullptr};
  bool col_stale = false;   // the retained forward ran the fused input block: colW1 / colW2 are rebuilt by backward()
  const float* bw_in[4] = {nullptr, nullptr, nullptr, nullptr};  // forward inputs, needed by the input-block wgrad
  static const int NWB = 7;
  bool grads_cleared = false;  // set by dbm_generator_step around backward(): cleargrads has just run (WgradBatch::cleared_target)
  int wbs_groups = -1;  // trunk groups the batches below were planned for
  WgradBatch wbs[NWB];  // batched weight gradients: tail, 5 trunk groups, pre-residual + input block (launched on the side stream)
  std::vector<DevBuf> cat, dA;
  DevBuf in_x, in_w1, in_w2, in_w3, a0, a3, a41, a42, off1, off2, col1, col2, a51, yout;
  DevBuf csr_ws;       // sampling lists of the deformable layers' input-gradient gather (deform_csr_build_kernel)
  // The lists depend on the layers' offsets only.  csr_early (set by dbm_train_iteration around the retained forward): forward() marks
  // the two offset tensors (ev_off), prebuild_csr(aux) builds both lists on `aux` beside the generator's loss (the 64 -> 64 layer's in
  // csr_ws, the 64 -> 1 layer's in csr_ws2) and backward() only waits for them (ev_csr) instead of building them on its own path.
  DevBuf csr_ws2;
  bool csr_early = false, csr_marked = false, csr_prebuilt = false;
  hipEvent_t ev_off[2] = {nullptr, nullptr}, ev_csr = nullptr;
  void prebuild_csr(hipStream_t aux);
  DevBuf dw2_partial;  // per-workgroup partial sums of final_conv_layer2's weight gradient (deform_bwd1_fused_kernel)
  bool deform_bwd_fused(int H4, int W4) const;
  bool deform_wgrad_fused(int H4, int W4) const;
  DevBuf dw1_partial;  // per-workgroup partial tiles of final_conv_layer1's weight gradient (deform_wgrad64_fused_kernel)
  DevBuf zdef;        // the last layer's premultiplied tap planes (N, 9 * out_ch, 4H, 4W): deform1_premul_kernel
  bool a42t_written = false;  // forward(): post_upsample_conv_layer_2 wrote the channels-last twin of its output itself
  bool zdef_kept = false;  // ... of the retained forward pass (the 64 -> 1 layer's backward in premultiplied form reads them)
  DevBuf gt2;         // backward: the transposed sampler applied to gy, (N, 9, 4H, 4W)
  DevBuf a42t, a51t;  // channels-last copies of the deformable layers' inputs (what the fused sampler gathers from)
  // bf16 sweep mode on large planes (conv_cl16.hip): the dense block's concat as NHWC bf16 (two buffers in ping-pong, 192
  // channels per pixel) and the 64-channel residual stream as NHWC fp32 (block input, block output, RRDB input)
  DevBuf catb[2], resb[4];
  DevBuf a3t, a41t;  // NHWC fp32 inputs of the two upsampling convolutions in the sweep's split-bf16 tail (conv_cl16x3_kernel)
  DevBuf a0t;        // the input block's 128-channel concat channels-last (the split-bf16 pre-residual convolution of the sweep)
  DevBuf a1t;        // NHWC fp32 copy of the trunk's input (the post-residual convolution's skip operand in that tail)
  DevBuf g_a0, g_a3, g_u1, g_z41, g_u2, g_a42, goff1, goff2, gcol, g_a51, g_y;
  // A second workspace on the same parameters: the G-step's generator forward can be enqueued while the D-step's
  // discriminator passes are still running (dbm_discriminator_step, prefetch flag).  The twin aliases this model's
  // arenas and packed weight images.  NOTE the library never has more than FOUR streams busy at once (main, side,
  // chain[0], chain[1]): with a fifth, streams share a hardware queue and independent chains serialise (measured:
  // every phase of the step 2x slower).
  Generator* twin = nullptr;
  Generator* owner = nullptr;  // twin only: the model whose arenas and weight images it aliases
  int chain_base = 0;
  bool use_aux = true;  // backward(): the deformable layers' offset-gradient kernel may run on chain[chain_base]
  bool wgrad_inline = false;  // backward(): every weight-gradient launch goes to the pass's OWN stream, behind the data-gradient chain
                              // (dbm_train_iteration, DBM_ITER_EARLY_TWIN=2: the side stream carries the discriminator's weight gradients, and
                              // a launch queued behind them would wait for the whole D-step)
  int max_split = 2;  // image ranges the 9x9 stage may be cut into (1: everything on the caller's stream)
  hipEvent_t ev_prefetch = nullptr;
  hipEvent_t ev_trunk = nullptr;   // forward(): recorded behind the 9x9 stage's trunk launch when mark_trunk is set (dbm_train_iteration:
  bool mark_trunk = false;         // the G-step's own forward may start there instead of behind this forward's full-resolution tail)
  hipEvent_t ev_pack[3] = {nullptr, nullptr, nullptr};  // pack_extra: main stream reached the repack / forward streams built / backward streams built
  // fused 9x9 trunk forward (trunk_fused.hip): per-wavefront weight streams (owner only), per-workspace hand-off granules
  float* tf_wstream = nullptr;
  float* tf_bstream = nullptr;
  float* tf_bwd_wstream = nullptr;   // transposed / tap-flipped streams of the fused data-gradient chain
  const float** tf_wsrc = nullptr;   // device tables of the trunk layers' W / b
  const float** tf_bsrc = nullptr;
  unsigned long long* tf_inbox = nullptr;
  int tf_epoch = 0;
  void pack_extra(hipStream_t s) override;
  bool trunk_fused_ok(int h, int w) const;
  Generator* get_twin();
  ~Generator() override;
  Generator(dbm_ctx* c, int n, float r, int oc);
  void ensure_ws(int N, int H, int W, bool train);
  int slot(int j) const { return wsTrain ? j : (j == 0 ? 0 : 1 + ((j - 1) & 3)); }
  void forward(int N, int H, int W, const float* x, const float* w1, const float* w2, const float* w3, float* y, bool keep);
  void backward(const float* gy);
};

struct Discriminator : dbm_model {
  int L_conv[10];  // 1..9 are igemm layers
  int T_c0W, T_c0b, T_bn[10][5];  // gamma, beta, avg_mean, avg_var, N
  int T_l1W, T_l1b, T_l2W, T_l2b;
  struct Cache {
    int N = 0, H = 0, W = 0;
    bool valid = false;
    DevBuf img, h[10], z[10], mean[10], istd[10], l1, out;
    const float* img_src = nullptr;   // the retained pass's input image: the private copy `img`, or the caller's buffer (borrow_images)
  } cache[3];         // [2]: the deferred eval-mode pass of dbm_train_iteration (never retained: no backward reads it)
  DevBuf bn_coef[3];  // eval-mode passes: [scale | shift] of all nine BatchNorm layers (launch_bn_eval_coeffs), one buffer per cache
                      // slot: two eval-mode passes in flight on different streams never share coefficients
  DevBuf g_h[2][2], g_z[2][10], g_l1[2], g_out, c0_scratch[2];  // per retained graph: the two backward passes overlap
  static const int NWG = 4;
  WgradBatch wbm[NWG];     // the same for BOTH graphs in one launch per group (the fused D-step: twice the work per launch)
  hipEvent_t ev_grp[2][NWG] = {};
  size_t comm_sent_lo = 0, comm_sent_hi = 0;  // gradient range already handed to the exchange by launch_group (this step)
  int merge_launcher = 0;    // merged mode: the slot whose backward pass is enqueued SECOND (it launches the groups)
  bool merge_slots = false;  // set by dbm_discriminator_step around its two backward calls (fake first, then real)
  // set by the fused steps around their retained forwards: conv_layer0's weight gradient reads the caller's image buffer directly (it
  // outlives the call, and the backward pass runs inside it) instead of a private copy -- one 332 KB copy launch less per pass (round 6)
  bool borrow_images = false;
  void launch_group(int slot, int g);
  WgradBatch wb[2][NWG];  // batched weight gradients per retained graph (real / fake batch): layers 9..6, 5..4, 3..2, 1
  Discriminator(dbm_ctx* c);
  void forward(int N, int H, int W, const float* img, float* logits, bool bn_train, bool keep, int slot, bool coef_ready = false);
  void prepare_eval_coeffs(int slot, hipStream_t s);
  void backward(int slot, const float* glogits, bool join = true);
};
