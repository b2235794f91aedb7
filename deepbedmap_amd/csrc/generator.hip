// GeneratorModel (reference srgan_train.py:421-576): parameter table, forward, backward.
#include "model.h"

static const float SLOPE = 0.2f;
static bool trunk_fused_enabled();

Generator::Generator(dbm_ctx* c, int n, float r, int oc) {
  ctx = c;
  type = 0;
  n_rrdb = n;
  rs = r;
  out_ch = oc;
  DBM_CHECK(n >= 1, "num_residual_blocks must be >= 1");
  // out_channels > 1 (srgan_train.py:450-457 allows it): forward only -- the reference's own training step fails its
  // F.mean_absolute_error shape check against the one-channel x_topo (:882-883) and the discriminator takes one channel
  DBM_CHECK(oc >= 1 && oc <= 16, "out_channels must be in [1, 16]");
  auto conv = [&](const std::string& name, int O, int C, int KH, int KW) {
    add_tensor(name + "/W", {O, C, KH, KW}, DBM_KIND_PARAM);
    add_tensor(name + "/b", {O}, DBM_KIND_PARAM);
  };
  // DeepbedmapInputBlock  srgan_train.py:223-254
  const char* in_names[4] = {"input_block/conv_on_X", "input_block/conv_on_W1", "input_block/conv_on_W2",
                             "input_block/conv_on_W3"};
  conv(in_names[0], 32, 1, 3, 3);
  conv(in_names[1], 32, 1, 30, 30);
  conv(in_names[2], 32, 2, 6, 6);
  conv(in_names[3], 32, 1, 3, 3);
  for (int i = 0; i < 4; ++i) {
    T_in[i][0] = tid(std::string(in_names[i]) + "/W");
    T_in[i][1] = tid(std::string(in_names[i]) + "/b");
  }
  L_in[1] = add_iglayer(in_names[1], 32, 1, 30, 1, 0, true, /*as_1x1=*/true);  // GEMM over the 900 im2col rows
  L_in[2] = add_iglayer(in_names[2], 32, 2, 6, 1, 0, true, /*as_1x1=*/true);   // 72 rows
  conv("pre_residual_conv_layer", 64, 128, 3, 3);  // :467-474
  L_pre = add_iglayer("pre_residual_conv_layer", 64, 128, 3, 1, 1, true);
  layers.back().want_x3 = true;
  const int cin[5] = {64, 96, 128, 160, 192}, cout[5] = {32, 32, 32, 32, 64};
  for (int i = 0; i < n; ++i)      // .repeat(n) -> Sequential children "0".."n-1"  :475-477
    for (int d = 1; d <= 3; ++d)   // ResInResDenseBlock  :383-391
      for (int k = 1; k <= 5; ++k) {  // ResidualDenseBlock  :292-331
        const std::string name = "residual_network/" + std::to_string(i) + "/residual_dense_block" + std::to_string(d) +
                                 "/conv_layer" + std::to_string(k);
        conv(name, cout[k - 1], cin[k - 1], 3, 3);
        L_rdb.push_back(add_iglayer(name, cout[k - 1], cin[k - 1], 3, 1, 1, true));
        layers.back().want_cl16 = true;
      }
  conv("post_residual_conv_layer", 64, 64, 3, 3);  // :478-485
  L_post = add_iglayer("post_residual_conv_layer", 64, 64, 3, 1, 1, true);
  layers.back().want_x3 = true;
  conv("post_upsample_conv_layer_1", 64, 64, 3, 3);  // :488-495
  L_up1 = add_iglayer("post_upsample_conv_layer_1", 64, 64, 3, 1, 1, true);
  layers.back().want_x3 = true;
  conv("post_upsample_conv_layer_2", 64, 64, 3, 3);  // :496-503
  L_up2 = add_iglayer("post_upsample_conv_layer_2", 64, 64, 3, 1, 1, true);
  layers.back().want_x3 = true;
  conv("final_conv_layer1/offset_conv", 18, 64, 3, 3);  // :506-514
  L_off1 = add_iglayer("final_conv_layer1/offset_conv", 18, 64, 3, 1, 1, true);
  layers.back().want_x3 = true;
  conv("final_conv_layer1/deform_conv", 64, 64, 3, 3);
  L_def1 = add_iglayer("final_conv_layer1/deform_conv", 64, 64, 3, 1, 0, true, /*as_1x1=*/true);
  layers.back().want_dx3 = true;
  conv("final_conv_layer2/offset_conv", 18, 64, 3, 3);  // :515-523
  L_off2 = add_iglayer("final_conv_layer2/offset_conv", 18, 64, 3, 1, 1, true);
  layers.back().want_x3 = true;
  conv("final_conv_layer2/deform_conv", oc, 64, 3, 3);
  T_def2W = tid("final_conv_layer2/deform_conv/W");
  T_def2b = tid("final_conv_layer2/deform_conv/b");
  // the persistent trunk kernels read their own weight streams: the trunk's per-layer images are built on demand only
  if (trunk_fused_enabled() && 3 * n + 1 <= TRUNK_FUSED_MAXCAT)
    for (int i : L_rdb) layers[i].lazy = true;
  alloc_arenas();
}

Generator::~Generator() {
  if (twin) delete twin;
  if (ev_prefetch) (void)hipEventDestroy(ev_prefetch);
  if (ev_trunk) (void)hipEventDestroy(ev_trunk);
  if (ev_csr) (void)hipEventDestroy(ev_csr);
  for (auto& e : ev_off) if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_pack) if (e) (void)hipEventDestroy(e);
  if (!is_view) {
    (void)hipFree(tf_wstream); (void)hipFree(tf_bstream); (void)hipFree(tf_bwd_wstream); (void)hipFree((void*)tf_wsrc); (void)hipFree((void*)tf_bsrc);
  }
  (void)hipFree(tf_inbox);
}

// ---- fused 9x9 trunk forward (trunk_fused.hip) ----
bool g_trunk_fused_off = false;
long g_step_serial = 0, g_trunk_rearm_at = -1;
static bool trunk_fused_enabled() {
  static const bool on = !(getenv("DBM_TRUNK_FUSED") && atoi(getenv("DBM_TRUNK_FUSED")) == 0);
  return on && !g_trunk_fused_off;
}

bool Generator::trunk_fused_ok(int h, int w) const {
  return h == 9 && w == 9 && !use_bf16 && 3 * n_rrdb + 1 <= TRUNK_FUSED_MAXCAT && trunk_fused_enabled();
}

void Generator::pack_extra(hipStream_t s) {
  if (is_view || !trunk_fused_enabled()) return;
  const int nrdb = 3 * n_rrdb;
  if (!tf_wstream) {
    const size_t nw = trunk_fused_stream_floats(nrdb);
    DBM_HIP(hipMalloc((void**)&tf_wstream, nw * sizeof(float)));
    DBM_HIP(hipMemset(tf_wstream, 0, nw * sizeof(float)));
    DBM_HIP(hipMalloc((void**)&tf_bstream, (size_t)nrdb * 192 * sizeof(float)));
    const size_t nb = trunk_fused_bwd_stream_floats(nrdb);
    DBM_HIP(hipMalloc((void**)&tf_bwd_wstream, nb * sizeof(float)));
    DBM_HIP(hipMemset(tf_bwd_wstream, 0, nb * sizeof(float)));
    std::vector<const float*> ws(nrdb * 5), bs(nrdb * 5);
    for (int i = 0; i < nrdb * 5; ++i) {
      ws[i] = P(layers[L_rdb[i]].wi);
      bs[i] = P(layers[L_rdb[i]].bi);
    }
    DBM_HIP(hipMalloc((void**)&tf_wsrc, ws.size() * sizeof(float*)));
    DBM_HIP(hipMalloc((void**)&tf_bsrc, bs.size() * sizeof(float*)));
    DBM_HIP(hipMemcpy((void*)tf_wsrc, ws.data(), ws.size() * sizeof(float*), hipMemcpyHostToDevice));
    DBM_HIP(hipMemcpy((void*)tf_bsrc, bs.data(), bs.size() * sizeof(float*), hipMemcpyHostToDevice));
    DBM_HIP(hipDeviceSynchronize());
  }
  // The per-layer images (launched by the caller on `s` just before) are needed by the very next kernels, the input block's;
  // the trunk's weight streams only by the persistent kernels behind it, the backward one not before the G-step: they are
  // rebuilt on chain[0] (idle at this point of a step) and the persistent launches wait for their events (~90 us off the
  // critical path of a training step).
  static const int aside = DBM_TUNE_GETENV("PACK_ASIDE") ? atoi(DBM_TUNE_GETENV("PACK_ASIDE")) : 1;
  hipStream_t ps = (aside && ctx->chain[0] && ctx->chain[0] != s) ? ctx->chain[0] : s;
  if (!ev_pack[0]) for (auto& e : ev_pack) DBM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (ps != s) {
    DBM_HIP(hipEventRecord(ev_pack[0], s));
    DBM_HIP(hipStreamWaitEvent(ps, ev_pack[0], 0));
  }
  launch_pack_trunk_fused(tf_wsrc, tf_bsrc, tf_wstream, tf_bstream, nrdb, ps);
  DBM_HIP(hipEventRecord(ev_pack[1], ps));
  launch_pack_trunk_fused_bwd(tf_wsrc, tf_bwd_wstream, nrdb, ps);
  DBM_HIP(hipEventRecord(ev_pack[2], ps));
}

Generator* Generator::get_twin() {
  if (twin) return twin;
  Generator* t = new Generator(ctx, n_rrdb, rs, out_ch);
  // drop what the constructor allocated and alias this model's parameters, gradients and packed weight images
  for (auto& L : t->layers) {
    if (L.wf) (void)hipFree(L.wf);
    for (int i = 0; i < 4; ++i)
      if (L.wb[i]) (void)hipFree(L.wb[i]);
  }
  (void)hipFree(t->params); (void)hipFree(t->grads); (void)hipFree(t->adam_m); (void)hipFree(t->adam_v); (void)hipFree(t->pers);
  (void)hipFree(t->d_adam_skipped);
  t->d_adam_skipped = nullptr;
  t->layers = layers;
  t->params = params; t->grads = grads; t->adam_m = adam_m; t->adam_v = adam_v; t->pers = pers;
  t->is_view = true;
  t->owner = this;
  t->packed_dirty = false;
  t->bwd_dirty = false;
  t->chain_base = 0;  // its main stream is chain[1] (set by the caller); the second image range shares chain[0]
  twin = t;
  return t;
}

// number of image ranges the 9x9 stage is cut into (1 or 2): only when a single range would leave the chip
// mostly idle (few tiles) and the ranges stay equal
static int trunk_split(int N, long hw) {
  static const int forced = DBM_TUNE_GETENV("TRUNK_SPLIT") ? atoi(DBM_TUNE_GETENV("TRUNK_SPLIT")) : 0;
  const long tiles = ((long)N * hw + 31) / 32;
  // two ranges: measured +4 % on the training step; more streams than hardware queues (four) serialise: 2.4x slower
  int ns = 1;
  if (tiles <= 1024 && N >= 2 && tiles >= 64) ns = 2;
  if (forced > 0 && forced <= 2 && N >= forced) ns = forced;
  return ns;
}

// The backward of the two deformable layers runs on the fused kernels (deform_fused.hip) when the forward did and the
// planes fit the CSR input-gradient kernel.
bool Generator::deform_bwd_fused(int H4, int W4) const {
  static const int fused_env = DBM_TUNE_GETENV("DEFORM_FUSED") ? atoi(DBM_TUNE_GETENV("DEFORM_FUSED")) : 1;
  return fused_env && out_ch == 1 && deform_conv_fused_ok(64, 64) && deform_input_grad_ok(64, H4, W4);
}

// final_conv_layer1's weight gradient from the channels-last input and the offsets (deform_wgrad64_fused_kernel) instead of from the retained
// sample matrix.  DBM_DEFORM_WGRAD_FUSED=0: the sample matrix + the 1x1 form (A/B; both are parity-tested).
bool Generator::deform_wgrad_fused(int H4, int W4) const {
  static const int env = getenv("DBM_DEFORM_WGRAD_FUSED") ? atoi(getenv("DBM_DEFORM_WGRAD_FUSED")) : 1;
  return env != 0 && deform_bwd_fused(H4, W4) && deform_conv_fused_ok(64, 64) && deform_conv_fused_ok(64, out_ch);
}

void Generator::ensure_ws(int N, int H, int W, bool train) {
  const bool same = (N == wsN && H == wsH && W == wsW);
  if (same && (wsTrain || !train)) return;
  DBM_CHECK(H >= 3 && W >= 3, "input tile must be at least 3x3");
  const size_t hw = (size_t)(H - 2) * (W - 2), n = (size_t)N;
  const int nrdb = 3 * n_rrdb;
  const bool tr = train || (same && wsTrain);
  in_x.ensure(n * H * W);
  in_w1.ensure(n * 100 * H * W);
  in_w2.ensure(n * 2 * 4 * H * W);
  in_w3.ensure(n * H * W);
  a0.ensure(n * 128 * hw);
  colW1.ensure(n * layers[L_in[1]].CinP * hw);
  colW2.ensure(n * layers[L_in[2]].CinP * hw);
  const int ncat = tr ? nrdb + 1 : 5;
  if ((int)cat.size() < ncat) cat.resize(ncat);
  for (int i = 0; i < ncat; ++i) cat[i].ensure(n * 192 * hw);
  a3.ensure(n * 64 * hw);
  a41.ensure(n * 64 * 4 * hw);
  a42.ensure(n * 64 * 16 * hw);
  off1.ensure(n * 32 * 16 * hw);
  off2.ensure(n * 32 * 16 * hw);
  a51.ensure(n * 64 * 16 * hw);
  a42t.ensure(n * 64 * 16 * hw);
  a51t.ensure(n * 64 * 16 * hw);
  yout.ensure(n * out_ch * 16 * hw);
  if (tr) {
    DBM_CHECK(out_ch == 1, "a retained (training) forward needs out_channels == 1: the reference's training step itself fails "
                           "with more (mean_absolute_error against the one-channel x_topo, srgan_train.py:882-883)");
    col1.ensure(n * 576 * 16 * hw);  // sample matrix of the 64 -> 64 deformable layer: retained passes only (its weight gradient)
    if ((int)dA.size() < nrdb + 1) dA.resize(nrdb + 1);
    for (int i = 0; i <= nrdb; ++i) dA[i].ensure(n * (i == nrdb ? 64 : 192) * hw);
    g_a0.ensure(n * 128 * hw);
    g_a3.ensure(n * 64 * hw);
    g_u1.ensure(n * 64 * 4 * hw);
    g_z41.ensure(n * 64 * 4 * hw);
    g_u2.ensure(n * 64 * 16 * hw);
    g_a42.ensure(n * 64 * 16 * hw);
    goff1.ensure(n * 32 * 16 * hw);
    goff2.ensure(n * 32 * 16 * hw);
    gcol.ensure(n * 576 * 16 * hw);
    g_a51.ensure(n * 64 * 16 * hw);
    g_y.ensure(n * 16 * hw);
  }
  if (!same) {
    // the 14 padding channels of the offset tensors must read as zero in the new layout
    hipStream_t s = ctx->stream;
    DBM_HIP(hipMemsetAsync(off1.p, 0, sizeof(float) * n * 32 * 16 * hw, s));
    DBM_HIP(hipMemsetAsync(off2.p, 0, sizeof(float) * n * 32 * 16 * hw, s));
    if (tr) {
      DBM_HIP(hipMemsetAsync(goff1.p, 0, sizeof(float) * n * 32 * 16 * hw, s));
      DBM_HIP(hipMemsetAsync(goff2.p, 0, sizeof(float) * n * 32 * 16 * hw, s));
    }
    have_graph = false;
  }
  if (!same || tr != wsTrain)
    for (auto& b : wbs) b.reset();  // buffers may have moved: re-plan the batched weight gradients
  wsN = N; wsH = H; wsW = W; wsTrain = tr;
}

void Generator::forward(int N, int H, int W, const float* x, const float* w1, const float* w2, const float* w3,
                        float* y, bool keep) {
  ensure_ws(N, H, W, keep);
  ensure_packed();
  // by-products of an EARLIER retained pass (sampling lists built ahead of its backward pass, premultiplied tap planes, the channels-last
  // twin) never survive into this one: a pass that threw between prebuild_csr and backward() would otherwise hand stale lists to the next
  csr_prebuilt = false; csr_marked = false; zdef_kept = false; a42t_written = false;
  hipStream_t s = ctx->stream;
  const int h = H - 2, w = W - 2;
  const long hw = (long)h * w;
  const int nrdb = 3 * n_rrdb;
  // bf16 sweep mode (DBM_BF16): layers that keep the fp32 arithmetic.  Bits: 1 input-block GEMMs, 2 pre- / post-residual
  // convs, 4 trunk, 8 upsampling convs, 16 offset convs of the deformable layers (the deformable GEMMs are always fp32);
  // 32 / 64: the post- / pre-residual conv in fp32 igemm instead of split-bf16 on the channels-last planes (round 5: bit 2 alone
  // now means "not in plain bf16" for them -- they run like the other signal-path layers, three bf16 MFMAs per product).
  // Default 27 = ONLY THE TRUNK multiplies in bf16 (83 % of the forward FLOPs).  Measured in metres at the reference's data
  // range (tools/bf16_error_study.py, DESIGN.md "bf16 at the data range"): the trunk's residual branches enter the signal
  // through two 0.1 scalings, its bf16 rounding costs 1 m rms of 2000 m of relief; every layer ON the signal path costs
  // 65-120 m rms in bf16 (all of them: 190 m), because the two deformable layers turn a 0.1 % error of their input and of
  // their sampling offsets into sampling-position errors on steep data.
  static const int bf16_keep32 = getenv("DBM_BF16_FP32_LAYERS") ? atoi(getenv("DBM_BF16_FP32_LAYERS")) : 27;
  auto prec = [&](ConvDesc d, int bit) {
    if (bf16_keep32 & bit) d.wp16 = nullptr;
    return d;
  };
  // Which trunk / tail forms this forward takes (decided up front: the bf16 sweep mode's layouts start at the input block).
  const bool fused = trunk_fused_ok(h, w);  // the whole trunk as one persistent launch (trunk_fused.hip)
  // bf16 sweep mode on planes the persistent kernels do not serve: the trunk on channels-last bf16 activations
  // (conv_cl16.hip).  DBM_CL16=0 (read per call): the per-layer implicit GEMM in its bf16 form instead.
  const bool cl16 = !fused && use_bf16 && !(bf16_keep32 & 4) && layers[L_rdb[0]].wcl16 != nullptr &&
                    !(getenv("DBM_CL16") && atoi(getenv("DBM_CL16")) == 0);
  // ... with the post-residual convolution and the full-resolution tail in split-bf16 on NHWC fp32 activations (see below)
  static const int fused_env_x3 = DBM_TUNE_GETENV("DEFORM_FUSED") ? atoi(DBM_TUNE_GETENV("DEFORM_FUSED")) : 1;
  const bool x3_tail = use_bf16 && !keep && fused_env_x3 && deform_conv_fused_ok(64, 64) && deform_conv_fused_ok(64, out_ch) &&
                       layers[L_up1].wx3 != nullptr && !(DBM_TUNE_GETENV("CL16X3") && atoi(DBM_TUNE_GETENV("CL16X3")) == 0);
  // DBM_POST_X3=0 / DBM_PRE_X3=0 (or bits 32 / 64 of DBM_BF16_FP32_LAYERS): the post- / pre-residual convolution in fp32 (igemm)
  // between layout conversions, as before round 5's last changes
  const bool post_x3 = cl16 && x3_tail && layers[L_post].wx3 != nullptr && !(bf16_keep32 & 32) &&
                       !(getenv("DBM_POST_X3") && atoi(getenv("DBM_POST_X3")) == 0);
  bool pre_x3 = post_x3 && layers[L_pre].wx3 != nullptr && !(bf16_keep32 & 64) && !(getenv("DBM_PRE_X3") && atoi(getenv("DBM_PRE_X3")) == 0);
  // ---- input block: four valid convolutions written straight into the 128-channel concat (:256-266) ----
  {
    struct { const float* in; int Cin, Hin, Win, K, stride; } br[4] = {
        {x, 1, H, W, 3, 1}, {w1, 1, 10 * H, 10 * W, 30, 10}, {w2, 2, 2 * H, 2 * W, 6, 2}, {w3, 1, H, W, 3, 1}};
    // training tile: one launch (input_block.hip); the wide branches' im2col images are then rebuilt by backward(), off this path.
    // DBM_INPUT_FUSED=0: layer by layer (A/B, and the form every other tile size takes)
    static const int in_fused_env = getenv("DBM_INPUT_FUSED") ? atoi(getenv("DBM_INPUT_FUSED")) : 1;
    const bool in_rows = in_fused_env && !input_block_fused_ok(H, W) && input_block_rows_ok(H, W) && !(use_bf16 && !(bf16_keep32 & 1));
    // (input_block_fused_kernel stages W1 / W2 with 16-byte loads and has no scalar form: 4-byte-aligned caller views take the layer-wise path)
    const bool in_al16 = (reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) % 16 == 0;
    const bool in_fused = in_rows || (in_fused_env && input_block_fused_ok(H, W) && in_al16 && !(use_bf16 && !(bf16_keep32 & 1)));
    col_stale = in_fused && keep;
    pre_x3 = pre_x3 && in_rows;   // (the split-bf16 pre-residual convolution reads the rows kernel's channels-last output)
    if (in_fused) {
      InputBlockLaunch q;
      q.yt = nullptr;
      if (pre_x3) {
        a0t.ensure((size_t)N * 128 * hw);
        q.yt = a0t.p;
      }
      q.x = x; q.w1 = w1; q.w2 = w2; q.w3 = w3;
      q.wx = P(T_in[0][0]); q.bx = P(T_in[0][1]);
      q.wf1 = layers[L_in[1]].wf; q.b1 = P(T_in[1][1]);
      q.wf2 = layers[L_in[2]].wf; q.b2 = P(T_in[2][1]);
      q.w3w = P(T_in[3][0]); q.b3 = P(T_in[3][1]);
      q.y = a0.p; q.ysn = 128 * hw; q.N = N;
      if (in_rows) launch_input_block_rows(q, H, W, s);
      else launch_input_block_fused(q, s);
    }
    for (int i = 0; i < 4 && !in_fused; ++i) {
      SmallConvDesc d;
      memset(&d, 0, sizeof(d));
      d.x = br[i].in; d.xsn = (long)br[i].Cin * br[i].Hin * br[i].Win; d.Cin = br[i].Cin; d.Hin = br[i].Hin; d.Win = br[i].Win;
      d.w = P(T_in[i][0]); d.bias = P(T_in[i][1]);
      d.y = a0.p + (long)i * 32 * hw; d.ysn = 128 * hw; d.Cout = 32; d.OH = h; d.OW = w;
      d.KH = d.KW = br[i].K; d.stride = br[i].stride; d.pad = 0; d.N = N; d.act = 0; d.slope = SLOPE;
      DBM_CHECK((br[i].Hin - br[i].K) / br[i].stride + 1 == h && (br[i].Win - br[i].K) / br[i].stride + 1 == w,
                "input block branch does not produce the (H-2, W-2) grid");
      if (L_in[i] >= 0) {  // wide kernels: im2col (0.1 % of the generator's bytes) + MFMA GEMM
        const IgLayer& L = layers[L_in[i]];
        float* col = (i == 1 ? colW1 : colW2).p;
        launch_im2col(br[i].in, col, N, br[i].Cin, br[i].Hin, br[i].Win, br[i].K, br[i].K, br[i].stride, h, w, L.CinP, s);
        ConvDesc g = prec(fwd_desc(L, col, (long)L.CinP * hw, h, w, 0, d.y, 128 * hw, N), 1);
        launch_igemm_conv(g, s);
      } else {
        launch_smallcin_conv_fwd(d, s);
      }
    }
  }
  // The 9x9 stage is a chain of ~180 short, latency-bound kernels (162-324 tiles each at batch 64).  The generator
  // has no cross-sample coupling, so the batch is cut into `nsplit` image ranges that run the same chain on
  // separate HIP streams: while one range's kernel is in its prologue / epilogue the other's feeds the MFMA pipes.
  DBM_MARK(s, "  gen_forward:input_block");
  const int nsplit = fused ? 1 : std::min(trunk_split(N, hw), max_split);
  auto cn0 = [&](int c) { return (long)(((long)c * N) / nsplit); };          // first image of range c
  auto cnc = [&](int c) { return (int)(cn0(c + 1) - cn0(c)); };                // images in range c
  auto cstream = [&](int c) { return c == 0 ? s : ctx->chain[chain_base + c - 1]; };
  for (int c = 1; c < nsplit; ++c) ctx->fork(s, cstream(c), c);
  // ---- pre-residual conv + LeakyReLU -> cat[0][:, :64]  (:541-542) ----
  if (pre_x3) {
    // bf16 sweep: split-bf16 on the channels-last concat, straight into the trunk's two operands (the fp32 residual stream and the bf16
    // concat's channels 0..63): no fp32 igemm launch, no nchw_to_cl
    for (auto& b : catb) b.ensure((size_t)N * 96 * hw);   // 192 bf16 = 96 floats per pixel
    for (auto& b : resb) b.ensure((size_t)N * 64 * hw);
    ClX3Launch q;
    memset(&q, 0, sizeof(q));
    q.x = a0t.p; q.xc = 128; q.Cin = 128; q.Cout = 64; q.ups = 0; q.w = layers[L_pre].wx3; q.bias = P(layers[L_pre].bi);
    q.y32 = resb[0].p; q.yc = 64; q.y16 = catb[0].p; q.y16c = 192; q.act = 1; q.slope = SLOPE; q.N = N; q.H = h; q.W = w;
    launch_conv_cl16x3(q, s);
  }
  for (int c = 0; c < (pre_x3 ? 0 : nsplit); ++c) {
    const long n0 = cn0(c);
    ConvDesc d = prec(fwd_desc(layers[L_pre], a0.p + n0 * 128 * hw, 128 * hw, h, w, 0, cat[0].p + n0 * 192 * hw, 192 * hw, cnc(c)), 2);
    d.act = 1;
    launch_igemm_conv(d, cstream(c));
  }
  // ---- RRDB trunk (:546; RDB :333-360, RRDB :393-404) ----
  if (fused) {
    const Generator* src = owner ? owner : this;
    DBM_CHECK(src->tf_wstream != nullptr, "fused trunk: weight streams not packed");
    const int IMGS = ctx->trunk_imgs;  // three workgroups per image, all resident at once: 64 images = 192 of 256 CUs
    if (!tf_inbox) {
      DBM_HIP(hipMalloc((void**)&tf_inbox, trunk_fused_inbox_bytes(64)));
      DBM_HIP(hipMemsetAsync(tf_inbox, 0, trunk_fused_inbox_bytes(64), s));
    }
    std::vector<float*> ptrs(nrdb + 1);
    for (int i = 0; i <= nrdb && keep; ++i) ptrs[i] = cat[i].p;
    for (int i0 = 0; i0 < N; i0 += IMGS) {
      TrunkFusedLaunch L;
      L.wstream = src->tf_wstream; L.bstream = src->tf_bstream; L.in = cat[0].p;
      L.cat = keep ? ptrs.data() : nullptr; L.out = cat[slot(nrdb)].p;
      L.inbox = tf_inbox; L.err = ctx->dev_err_d; L.err_dev = ctx->dev_err_flag;
      L.nrdb = nrdb; L.nimg = std::min(IMGS, N - i0); L.img0 = i0; L.epoch = ++tf_epoch;
      L.rs = rs; L.slope = SLOPE;
      // data-parallel: at most 192 resident workgroups, so that RCCL's kernels (and everything else) keep 64 compute units
      L.no_helper = ctx->comm_active() ? 1 : 0;
      if (src->ev_pack[1]) DBM_HIP(hipStreamWaitEvent(s, src->ev_pack[1], 0));  // the weight streams (pack_extra)
      ctx->persist_begin(s);
      launch_trunk_fused(L, s);
      ctx->persist_end(s);
    }
    if (mark_trunk) {
      if (!ev_trunk) DBM_HIP(hipEventCreateWithFlags(&ev_trunk, hipEventDisableTiming));
      DBM_HIP(hipEventRecord(ev_trunk, s));
    }
  }
  if (cl16) {
    const size_t n = (size_t)N;
    for (auto& b : catb) b.ensure(n * 96 * hw);   // 192 bf16 = 96 floats per pixel
    for (auto& b : resb) b.ensure(n * 64 * hw);
    for (int c = 1; c < nsplit; ++c) ctx->fork(cstream(c), s, 4 + c);  // (the pre-residual conv's image ranges)
    if (!pre_x3) launch_nchw_to_cl(cat[0].p, 192 * hw, resb[0].p, catb[0].p, 192, N, (int)hw, s);
    if (post_x3) {   // (resb[0] is recycled by the fourth dense block: the skip operand of the post-residual convolution keeps its own copy)
      a1t.ensure(n * 64 * hw);
      DBM_HIP(hipMemcpyAsync(a1t.p, resb[0].p, sizeof(float) * n * 64 * hw, hipMemcpyDeviceToDevice, s));
    }
    for (int j = 0; j < nrdb; ++j) {
      void* C16 = catb[j & 1].p;
      for (int k = 0; k < 5; ++k) {
        const IgLayer& L = layers[L_rdb[j * 5 + k]];
        ClConvLaunch q;
        memset(&q, 0, sizeof(q));
        q.x = C16; q.xc = 192; q.Cin = 64 + 32 * k; q.Cout = k < 4 ? 32 : 64; q.w = L.wcl16; q.bias = P(L.bi);
        q.N = N; q.H = h; q.W = w; q.slope = SLOPE; q.s1 = 1.f; q.s2 = 1.f; q.zeros = ctx->zeros;
        if (k < 4) {
          q.y16 = C16; q.yc = 192; q.y0 = q.Cin; q.act = 1;
        } else {
          q.y16 = catb[(j + 1) & 1].p; q.yc = 192; q.y0 = 0;
          q.y32 = resb[(j + 1) & 3].p;
          q.r1 = resb[j & 3].p; q.s1 = rs;                                    // a6 = a5*rs + a0  (:358)
          if (j % 3 == 2) { q.r2 = resb[(3 * (j / 3)) & 3].p; q.s2 = rs; }    // a4 = a3*rs + x   (:402)
        }
        launch_conv_cl16(q, s);
      }
    }
    if (post_x3) {
      // a3 = a1 + conv(a2) (:550-551) straight from / to NHWC fp32 in split-bf16: no cl_to_nchw, no fp32 igemm launch, no nchw_to_cl
      a3t.ensure(n * 64 * hw);
      ClX3Launch q;
      memset(&q, 0, sizeof(q));
      q.x = resb[nrdb & 3].p; q.xc = 64; q.Cin = 64; q.Cout = 64; q.ups = 0; q.w = layers[L_post].wx3; q.bias = P(layers[L_post].bi);
      q.y32 = a3t.p; q.yc = 64; q.r1 = a1t.p; q.r1c = 64; q.act = 0; q.slope = SLOPE; q.N = N; q.H = h; q.W = w;
      launch_conv_cl16x3(q, s);
    } else {
      launch_cl_to_nchw(resb[nrdb & 3].p, cat[slot(nrdb)].p, 192 * hw, N, (int)hw, s);
    }
    for (int c = 1; c < nsplit; ++c) ctx->fork(s, cstream(c), c);  // the post-residual conv's ranges continue behind it
  }
  if (!fused && !cl16) (owner ? owner : this)->ensure_packed_lazy();
  for (int j = 0; j < ((fused || cl16) ? 0 : nrdb); ++j) {
    for (int c = 0; c < nsplit; ++c) {  // one dense block per range at a time: fewer stream switches on the host
      for (int k = 0; k < 5; ++k) {
        const long n0 = cn0(c) * 192 * hw;
        const int Nc = cnc(c);
        float* C = cat[slot(j)].p + n0;
        if (k < 4) {
          const int cin = 64 + 32 * k;
          ConvDesc d = prec(fwd_desc(layers[L_rdb[j * 5 + k]], C, 192 * hw, h, w, 0, C + (long)cin * hw, 192 * hw, Nc), 4);
          d.act = 1;
          launch_igemm_conv(d, cstream(c));
        } else {
          float* Cn = cat[slot(j + 1)].p + n0;
          ConvDesc d = prec(fwd_desc(layers[L_rdb[j * 5 + 4]], C, 192 * hw, h, w, 0, Cn, 192 * hw, Nc), 4);
          d.s1 = rs; d.r1 = C; d.r1sn = 192 * hw; d.r1_nch = 64; d.r1s = 1.f;  // a6 = a5*rs + a0  (:358)
          if (j % 3 == 2) {  // a4 = a3*rs + x  (:402)
            d.r2 = cat[slot(j - 2)].p + n0; d.r2sn = 192 * hw; d.s2 = rs;
          }
          launch_igemm_conv(d, cstream(c));
        }
      }
    }
  }
  // ---- post-residual conv, a3 = a1 + conv(a2)  (:550-551) ----
  for (int c = 0; c < (post_x3 ? 0 : nsplit); ++c) {
    const long n0 = cn0(c);
    ConvDesc d = prec(fwd_desc(layers[L_post], cat[slot(nrdb)].p + n0 * 192 * hw, 192 * hw, h, w, 0, a3.p + n0 * 64 * hw, 64 * hw, cnc(c)), 2);
    d.r1 = cat[0].p + n0 * 192 * hw; d.r1sn = 192 * hw; d.r1_nch = 64;
    launch_igemm_conv(d, cstream(c));
  }
  for (int c = 1; c < nsplit; ++c) ctx->fork(cstream(c), s, 4 + c);
  DBM_MARK(s, "  gen_forward:9x9_stage");
  // ---- nearest x2 + conv + LeakyReLU, twice; the resize is folded into the conv's gather (:556-568) ----
  const int H4 = 4 * h, W4 = 4 * w;
  const long P4 = 16 * hw;
  // The sampler is fused into the GEMM (deform_fused.hip), fed from a channels-last copy of the layer input; the
  // (N, 576, H, W) sample matrices exist only in a retained pass, as a by-product for the two weight gradients.
  static const int fused_env = DBM_TUNE_GETENV("DEFORM_FUSED") ? atoi(DBM_TUNE_GETENV("DEFORM_FUSED")) : 1;
  const bool dfused = fused_env && deform_conv_fused_ok(64, 64) && deform_conv_fused_ok(64, out_ch);
  // bf16 sweep mode: the upsampling and offset convolutions -- on the signal path, where plain bf16 costs ~100 m rms at the
  // data range -- run in split-bf16 arithmetic (conv_cl16x3_kernel: three bf16 MFMAs per product, 2^-16 operand precision) on
  // NHWC fp32 activations, which is also what the fused deformable sampler reads.  DBM_CL16X3=0 (read per call): fp32 igemm.
  const bool x3 = use_bf16 && !keep && dfused && layers[L_up1].wx3 != nullptr && !(DBM_TUNE_GETENV("CL16X3") && atoi(DBM_TUNE_GETENV("CL16X3")) == 0);
  auto x3_launch = [&](const IgLayer& L, const float* xin, int ups, int Ho, int Wo, float* y32, float* yp, int act) {
    ClX3Launch q;
    memset(&q, 0, sizeof(q));
    q.x = xin; q.xc = 64; q.Cin = 64; q.Cout = L.O; q.ups = ups; q.w = L.wx3; q.bias = P(L.bi);
    q.y32 = y32; q.yc = 64; q.yp = yp; q.ysn = 32L * Ho * Wo; q.ypc = L.O; q.act = act; q.slope = SLOPE; q.N = N; q.H = Ho; q.W = Wo;
    launch_conv_cl16x3(q, s);
  };
  if (x3) {
    a3t.ensure((size_t)N * 64 * hw);
    a41t.ensure((size_t)N * 64 * 4 * hw);
    a42t.ensure((size_t)N * 64 * P4);
    a51t.ensure((size_t)N * 64 * P4);
    if (!post_x3) launch_nchw_to_cl(a3.p, 64 * hw, a3t.p, nullptr, 0, N, (int)hw, s);
    x3_launch(layers[L_up1], a3t.p, 1, 2 * h, 2 * w, a41t.p, nullptr, 1);
    x3_launch(layers[L_up2], a41t.p, 1, H4, W4, a42t.p, nullptr, 1);
  } else {
    ConvDesc d = prec(fwd_desc(layers[L_up1], a3.p, 64 * hw, h, w, 1, a41.p, 64 * 4 * hw, N), 8);
    d.act = 1;
    launch_igemm_conv(d, s);
    ConvDesc e = prec(fwd_desc(layers[L_up2], a41.p, 64 * 4 * hw, 2 * h, 2 * w, 1, a42.p, 64 * 16 * hw, N), 8);
    e.act = 1;
    // (the fused deformable sampler reads a channels-last copy of this output: the LDS-tiled form writes it from its epilogue)
    static const int fused_env0 = DBM_TUNE_GETENV("DEFORM_FUSED") ? atoi(DBM_TUNE_GETENV("DEFORM_FUSED")) : 1;
    a42t_written = false;
    if (fused_env0 && deform_conv_fused_ok(64, 64) && deform_conv_fused_ok(64, out_ch)) {
      a42t.ensure((size_t)N * 64 * 16 * hw);
      e.yt = a42t.p;
      a42t_written = conv_tile_writes_yt(e);
      if (!a42t_written) e.yt = nullptr;
    }
    launch_igemm_conv(e, s);
  }
  // ---- deformable conv 1 + LeakyReLU (:572-573): offset conv, sampler -> col, GEMM over 576 columns ----
  if (dfused) {
    a42t.ensure((size_t)N * 64 * P4);
    a51t.ensure((size_t)N * 64 * P4);
  } else {
    col1.ensure((size_t)N * 576 * P4);
    if (keep) col2.ensure((size_t)N * 576 * P4);
  }
  if (x3) {
    x3_launch(layers[L_off1], a42t.p, 0, H4, W4, nullptr, off1.p, 0);
    // (the split-bf16 tail reads channels-last only: the NCHW copy of this layer's output is not written)
    static const bool dx3 = !(DBM_TUNE_GETENV("DEFORM_X3") && atoi(DBM_TUNE_GETENV("DEFORM_X3")) == 0);
    if (dx3 && layers[L_def1].wdx3)
      launch_deform_conv64_x3(a42t.p, off1.p, layers[L_def1].wdx3, P(layers[L_def1].bi), nullptr, a51t.p, N, H4, W4, 32 * P4, 1, SLOPE, s);
    else
      launch_deform_conv_fused(a42t.p, off1.p, layers[L_def1].wf, P(layers[L_def1].bi), nullptr, a51t.p, nullptr, N, 64, H4, W4, 32 * P4, 64,
                               1, SLOPE, s);
  } else {
    ConvDesc d = prec(fwd_desc(layers[L_off1], a42.p, 64 * P4, H4, W4, 0, off1.p, 32 * P4, N), 16);
    launch_igemm_conv(d, s);
    csr_marked = false;
    if (keep && csr_early) {
      for (auto& e : ev_off)
        if (!e) DBM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      DBM_HIP(hipEventRecord(ev_off[0], s));
    }
    if (dfused) {
      if (!a42t_written) launch_nchw_to_nhwc64(a42.p, a42t.p, N, (int)P4, s);
      // (round 6: the layer's weight gradient re-samples -- deform_wgrad64_fused_kernel -- so a retained pass no longer writes the
      //  191 MB sample matrix from this kernel's tap loop; only the unfused backward path still reads it)
      float* colout = (keep && !deform_wgrad_fused(H4, W4)) ? col1.p : nullptr;
      launch_deform_conv_fused(a42t.p, off1.p, layers[L_def1].wf, P(layers[L_def1].bi), a51.p, a51t.p, colout, N, 64, H4,
                               W4, 32 * P4, 64, 1, SLOPE, s);
    } else {
      launch_deform_sample(a42.p, off1.p, col1.p, N, 64, H4, W4, 32 * P4, s);
      ConvDesc g = fwd_desc(layers[L_def1], col1.p, 576 * P4, H4, W4, 0, a51.p, 64 * P4, N);
      g.act = 1;
      launch_igemm_conv(g, s);
    }
  }
  // ---- deformable conv 2 (:574) ----
  {
    if (x3) {
      x3_launch(layers[L_off2], a51t.p, 0, H4, W4, nullptr, off2.p, 0);
    } else {
      ConvDesc d = prec(fwd_desc(layers[L_off2], a51.p, 64 * P4, H4, W4, 0, off2.p, 32 * P4, N), 16);
      launch_igemm_conv(d, s);
      if (keep && csr_early) {
        DBM_HIP(hipEventRecord(ev_off[1], s));
        csr_marked = true;
      }
    }
    if (dfused) {
      static const bool premul = !(getenv("DBM_DEFORM1_PREMUL") && atoi(getenv("DBM_DEFORM1_PREMUL")) == 0);
      if (premul) zdef.ensure((size_t)N * 9 * out_ch * P4);
      launch_deform_conv_fused(a51t.p, off2.p, P(T_def2W), P(T_def2b), y, nullptr, nullptr, N, 64, H4, W4, 32 * P4, out_ch, 0, SLOPE, s,
                               premul ? zdef.p : nullptr);
      zdef_kept = premul && keep && out_ch == 1;
      // (unfused backward only: the 64 -> 1 layer's weight gradient then reads its sample matrix)
      if (keep && !deform_bwd_fused(H4, W4)) {
        col2.ensure((size_t)N * 576 * P4);
        launch_deform_sample(a51.p, off2.p, col2.p, N, 64, H4, W4, 32 * P4, s);
      }
    } else {
      DBM_CHECK(out_ch == 1, "the unfused deformable tail serves out_channels == 1 only");
      float* col = keep ? col2.p : col1.p;
      launch_deform_sample(a51.p, off2.p, col, N, 64, H4, W4, 32 * P4, s);
      launch_gemv_cols(col, P(T_def2W), P(T_def2b), y, N, 576, (int)P4, s);
    }
  }
  bw_in[0] = x; bw_in[1] = w1; bw_in[2] = w2; bw_in[3] = w3;
  have_graph = keep;
}

void Generator::prebuild_csr(hipStream_t aux) {
  if (!csr_marked || !have_graph) return;
  csr_marked = false;
  const int N = wsN, H4 = 4 * (wsH - 2), W4 = 4 * (wsW - 2);
  const long P4 = (long)H4 * W4;
  if (!deform_bwd_fused(H4, W4) || !deform_csr_lists_ok(64, H4, W4)) return;
  csr_ws.ensure(deform_csr_workspace_floats(N, H4, W4));
  csr_ws2.ensure(deform_csr_workspace_floats(N, H4, W4));
  if (!ev_csr) DBM_HIP(hipEventCreateWithFlags(&ev_csr, hipEventDisableTiming));
  DBM_HIP(hipStreamWaitEvent(aux, ev_off[0], 0));
  launch_deform_csr_build(off1.p, csr_ws.p, N, H4, W4, 32 * P4, aux);
  DBM_HIP(hipStreamWaitEvent(aux, ev_off[1], 0));
  launch_deform_csr_build(off2.p, csr_ws2.p, N, H4, W4, 32 * P4, aux);
  DBM_HIP(hipEventRecord(ev_csr, aux));
  csr_prebuilt = true;
}

void Generator::backward(const float* gy) {
  DBM_CHECK(have_graph && wsTrain, "generator backward without a retained forward (DBM_KEEP_GRAPH)");
  mark_grads_touched();
  hipStream_t s = ctx->stream;
  (owner ? owner : this)->ensure_packed_bwd();   // (a no-op inside dbm_train_iteration, which rebuilds them at its head)
  const int N = wsN, H = wsH, W = wsW, h = H - 2, w = W - 2;
  const long hw = (long)h * w, P4 = 16 * hw;
  const int H4 = 4 * h, W4 = 4 * w, nrdb = 3 * n_rrdb;
  for (auto& b : wbs) b.cleared_target = grads_cleared;
  // ---- final_conv_layer2 (deformable, 64 -> 1) ----
  const bool bfused = deform_bwd_fused(H4, W4);
  const bool pre_csr = csr_prebuilt && bfused;   // (prebuild_csr: both layers' sampling lists are already being built on another stream)
  csr_prebuilt = false;
  if (bfused) {
    // offset gradients + the layer's weight / bias gradient from one pass over the channels-last input (no sample matrix);
    // on the aux stream next to the input-gradient gather when the caller has one
    hipStream_t sg = s;
    if (use_aux) {
      ctx->fork(s, ctx->chain[chain_base], 2);
      sg = ctx->chain[chain_base];
    }
    dw2_partial.ensure(deform_bwd1_partial_floats(N, H4, W4));
    csr_ws.ensure(deform_csr_workspace_floats(N, H4, W4));
    if (pre_csr) DBM_HIP(hipStreamWaitEvent(s, ev_csr, 0));
    DBM_MARK(s, "G:backward_begin");   // (behind the wait for the prebuilt sampling lists)
    // Round 5: in the premultiplied form of the forward pass (z_t = sum_c w[c][t] x_c kept from it) the layer's whole backward is a
    // CSR gather of ONE value per list entry, four single-float gathers per (position, tap) and one pass over the input -- instead of
    // gathering 9 x 4 x 256 bytes per position for the offset / weight gradients (150 us) and 64 values per entry for the input gradient.
    // DBM_DEFORM1_PREMUL_BWD=0: the gathering kernels (A/B).
    static const bool premul_bwd = !(getenv("DBM_DEFORM1_PREMUL_BWD") && atoi(getenv("DBM_DEFORM1_PREMUL_BWD")) == 0);
    if (premul_bwd && zdef_kept) {
      gt2.ensure((size_t)N * 9 * P4);
      launch_deform_bwd1_premul(a51t.p, off2.p, P(T_def2W), gy, zdef.p, goff2.p, g_a51.p, G(T_def2W), G(T_def2b), dw2_partial.p,
                                pre_csr ? csr_ws2.p : csr_ws.p, gt2.p, N, H4, W4, 32 * P4, s, pre_csr);
    } else {
      launch_deform_bwd1_fused(a51t.p, off2.p, P(T_def2W), gy, goff2.p, G(T_def2W), G(T_def2b), dw2_partial.p, N, H4, W4, 32 * P4, sg);
      launch_deform_input_grad(a51.p, off2.p, nullptr, P(T_def2W), gy, g_a51.p, N, 64, H4, W4, 32 * P4, s, pre_csr ? csr_ws2.p : csr_ws.p,
                               pre_csr);
    }
    if (sg != s) ctx->fork(sg, s, 3);
  } else {
    // (its weight gradient only needs gy and the retained columns: side stream, underneath the sampler's backward)
    ctx->fork_to_side(5);
    launch_gemv_cols_wgrad(col2.p, gy, G(T_def2W), G(T_def2b), N, 576, (int)P4, ctx->side);
    launch_deform_backward(a51.p, off2.p, nullptr, P(T_def2W), gy, g_a51.p, goff2.p, N, 64, H4, W4, 32 * P4, s,
                           use_aux ? ctx->chain[chain_base] : nullptr, use_aux ? &ctx->ev_fork[2] : nullptr);
  }
  {
    const IgLayer& L = layers[L_off2];
    run_wgrad(L, a51.p, 64 * P4, H4, W4, 0, goff2.p, 32 * P4, H4, W4, N, 1.f, &wbs[0]);
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = goff2.p; d.xsn = 32 * P4; d.N = N;
    d.y = g_a51.p; d.ysn = 64 * P4; d.accumulate = 1; d.s1 = 1.f; d.s2 = 1.f;
    d.mask = a51.p; d.masksn = 64 * P4; d.mask_c0 = 0;  // through F.leaky_relu (:573)
    run_dgrad(L, d, H4, W4);
  }
  // ---- final_conv_layer1 (deformable, 64 -> 64): g_a51 now holds d loss / d (pre-activation) ----
  {
    const IgLayer& L = layers[L_def1];
    // (its weight gradient: sampler-fused, launched with the tail's batch on the side stream below -- deform_wgrad_fused; else from the
    //  retained sample matrix through the batched 1x1 form)
    if (!deform_wgrad_fused(H4, W4)) run_wgrad(L, col1.p, 576 * P4, H4, W4, 0, g_a51.p, 64 * P4, H4, W4, N, 1.f, &wbs[0]);
    if (bfused) {
      // column gradients W^T gy on the MFMAs, offset gradients from the same LDS tile; then the input-gradient gather
      launch_deform_bwd64_fused(a42t.p, off1.p, L.wb[1], g_a51.p, gcol.p, goff1.p, N, H4, W4, 32 * P4, s);
      launch_deform_input_grad(a42.p, off1.p, gcol.p, nullptr, nullptr, g_a42.p, N, 64, H4, W4, 32 * P4, s, csr_ws.p, pre_csr);
    } else {
      ConvDesc d;
      memset(&d, 0, sizeof(d));
      d.x = g_a51.p; d.xsn = 64 * P4; d.N = N;
      d.y = gcol.p; d.ysn = 576 * P4; d.s1 = 1.f; d.s2 = 1.f;
      run_dgrad(L, d, H4, W4);
      launch_deform_backward(a42.p, off1.p, gcol.p, nullptr, nullptr, g_a42.p, goff1.p, N, 64, H4, W4, 32 * P4, s,
                             use_aux ? ctx->chain[chain_base] : nullptr, use_aux ? &ctx->ev_fork[2] : nullptr);
    }
  }
  {
    const IgLayer& L = layers[L_off1];
    run_wgrad(L, a42.p, 64 * P4, H4, W4, 0, goff1.p, 32 * P4, H4, W4, N, 1.f, &wbs[0]);
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = goff1.p; d.xsn = 32 * P4; d.N = N;
    d.y = g_a42.p; d.ysn = 64 * P4; d.accumulate = 1; d.s1 = 1.f; d.s2 = 1.f;
    d.mask = a42.p; d.masksn = 64 * P4; d.mask_c0 = 0;  // through F.leaky_relu (:568)
    run_dgrad(L, d, H4, W4);
  }
  // ---- post_upsample_conv_layer_2 on resize(a41) ----
  {
    const IgLayer& L = layers[L_up2];
    run_wgrad(L, a41.p, 64 * 4 * hw, 2 * h, 2 * w, 1, g_a42.p, 64 * P4, H4, W4, N, 1.f, &wbs[0]);
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = g_a42.p; d.xsn = 64 * P4; d.N = N;
    d.y = g_u2.p; d.ysn = 64 * P4; d.s1 = 1.f; d.s2 = 1.f;
    run_dgrad(L, d, H4, W4);
    launch_sumpool2(g_u2.p, a41.p, g_z41.p, (long)N * 64, 2 * h, 2 * w, SLOPE, s);  // resize bwd + lrelu' (:560)
  }
  // ---- post_upsample_conv_layer_1 on resize(a3) ----
  {
    const IgLayer& L = layers[L_up1];
    run_wgrad(L, a3.p, 64 * hw, h, w, 1, g_z41.p, 64 * 4 * hw, 2 * h, 2 * w, N, 1.f, &wbs[0]);
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = g_z41.p; d.xsn = 64 * 4 * hw; d.N = N;
    d.y = g_u1.p; d.ysn = 64 * 4 * hw; d.s1 = 1.f; d.s2 = 1.f;
    run_dgrad(L, d, 2 * h, 2 * w);
    launch_sumpool2(g_u1.p, nullptr, g_a3.p, (long)N * 64, h, w, SLOPE, s);
  }
  // ---- post_residual_conv_layer: a3 = a1 + conv(a2) ----
  {
    const IgLayer& L = layers[L_post];
    run_wgrad(L, cat[nrdb].p, 192 * hw, h, w, 0, g_a3.p, 64 * hw, h, w, N, 1.f, &wbs[0]);
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = g_a3.p; d.xsn = 64 * hw; d.N = N;
    d.y = dA[nrdb].p; d.ysn = 64 * hw; d.s1 = 1.f; d.s2 = 1.f;
    run_dgrad(L, d, h, w);
  }
  // Weight gradients never feed the data-gradient chain, and that chain (one short, latency-bound kernel per conv)
  // leaves most of the chip idle: the batches go to the side stream as soon as their inputs are final.
  DBM_MARK(s, "G:backward_tail_layers");
  ctx->fork_to_side(0);
  static const int iter_abl = DBM_MEASURE_ENV("ITER_ABL");  // (libdbm_measure.so only: 2 = no trunk weight gradients, 4 = none of the tail's)
  const bool inline_wg = wgrad_inline && !ctx->comm_in_step;
  if (!(iter_abl & 4) && !inline_wg) wbs[0].launch(ctx->side);
  if (deform_wgrad_fused(H4, W4) && !(iter_abl & 4)) {   // final_conv_layer1's weight / bias gradient (g_a51 and the offsets are final)
    dw1_partial.ensure(deform_wgrad64_partial_floats(N, H4, W4));
    launch_deform_wgrad64_fused(a42t.p, off1.p, g_a51.p, G(layers[L_def1].wi), G(layers[L_def1].bi), dw1_partial.p, N, H4, W4, 32 * P4,
                                inline_wg ? s : ctx->side);
  }
  if (col_stale) {  // (fused input block: the im2col images the wide branches' weight gradients read -- wbs[6], launched last)
    hipStream_t cs = inline_wg ? s : ctx->side;
    launch_im2col(bw_in[1] ? bw_in[1] : in_w1.p, colW1.p, N, 1, 10 * H, 10 * W, 30, 30, 10, h, w, layers[L_in[1]].CinP, cs);
    launch_im2col(bw_in[2] ? bw_in[2] : in_w2.p, colW2.p, N, 2, 2 * H, 2 * W, 6, 6, 2, h, w, layers[L_in[2]].CinP, cs);
    col_stale = false;
  }
  // Data-parallel run: the gradient arena is in construction order (input block | pre | trunk | tail), and the backward
  // pass finishes it from the end: every group of weight gradients that has been enqueued on the side stream is a
  // contiguous range that can be summed over ranks (chain[1]) while the rest of the pass still runs.
  auto rdb_off = [&](int j) { return j >= nrdb ? tensors[layers[L_post].wi].off : tensors[layers[L_rdb[j * 5]].wi].off; };
  if (ctx->comm_in_step) ctx->comm_bucket(grads + rdb_off(nrdb), nparam - rdb_off(nrdb), ctx->side);
  // ---- trunk, last dense block first ----
  const bool fused = trunk_fused_ok(h, w) && !(getenv("DBM_TRUNK_FUSED_BWD") && atoi(getenv("DBM_TRUNK_FUSED_BWD")) == 0);
  int final_hi = nrdb;  // dense blocks [0, final_hi): the trunk group whose weight gradients go out last, with the input block's
  // fused chain: ONE launch, the whole trunk's weight gradients behind it (round 2, after the chain and the weight-gradient
  // kernels got faster: 1: 8.85, 2: 9.04, 3: 9.00, 4: 9.05 ms per step; round 1: 12.19 / 12.14 / 12.23 / 12.17)
  // (data-parallel: four groups, so that the last, exposed, gradient bucket is a quarter of the trunk instead of all of it)
  static const int ngroups_forced = getenv("DBM_BWD_GROUPS") ? atoi(getenv("DBM_BWD_GROUPS")) : -1;
  const int ngroups_env = ngroups_forced >= 0 ? ngroups_forced : (ctx->comm_in_step ? 4 : 1);
  // (the batches' membership depends on the trunk PATH as well -- fused: one group per launch; layer-wise: the fixed five groups --
  //  and WgradBatch::add is a no-op once a batch is built: a path change (persistent kernels paused after a time-out, re-armed later)
  //  re-plans them, or the stale fused batch would be launched next to the layer-wise groups and count 11 of 12 RRDBs twice)
  const int wbs_key = fused ? ngroups_env : -2;
  if (wbs_key != wbs_groups) {
    for (int i = 1; i <= 5; ++i) wbs[i].reset();
    wbs_groups = wbs_key;
  }
  auto group_of = [&](int j) {
    // groups of residual-in-residual blocks, shrinking towards the end of the chain: what is still to do once the
    // data-gradient chain has finished (the last group + the pre-residual / input-block batch) is exposed time
    const int r = j / 3;
    if (fused && ngroups_env > 0) {  // k equal groups, numbered 5-k+1 .. 5 from the top of the trunk down
      const int k = std::min(std::min(ngroups_env, 5), n_rrdb);
      return 5 - (r * k) / n_rrdb;
    }
    return r == 0 ? 5 : r == 1 ? 4 : r <= 3 ? 3 : (r >= 4 + (n_rrdb - 3) / 2 ? 1 : 2);
  };
  if (fused) {  // one persistent launch per group (trunk_fused_bwd.hip); the group's weight gradients follow on the side stream
    const Generator* src = owner ? owner : this;
    DBM_CHECK(src->tf_bwd_wstream != nullptr, "fused trunk: weight streams not packed");
    const int IMGS = ctx->trunk_imgs;
    if (!tf_inbox) {
      DBM_HIP(hipMalloc((void**)&tf_inbox, trunk_fused_inbox_bytes(64)));
      DBM_HIP(hipMemsetAsync(tf_inbox, 0, trunk_fused_inbox_bytes(64), s));
    }
    std::vector<float*> dAp(nrdb);
    std::vector<const float*> catp(nrdb);
    for (int i = 0; i < nrdb; ++i) { dAp[i] = dA[i].p; catp[i] = cat[i].p; }
    int prev = -1, prev_lo = 0, prev_hi = 0;
    for (int j = nrdb - 1; j >= 0; --j) {
      const int grp = group_of(j);
      if (grp != prev) {
        if (prev >= 0) {
          ctx->fork_to_side(prev);
          wbs[prev].launch(ctx->side);
          if (ctx->comm_in_step) ctx->comm_bucket(grads + rdb_off(prev_lo), rdb_off(prev_hi) - rdb_off(prev_lo), ctx->side);
        }
        int jlo = j;
        while (jlo > 0 && group_of(jlo - 1) == grp) --jlo;
        for (int i0 = 0; i0 < N; i0 += IMGS) {
          TrunkFusedBwdLaunch L;
          L.wstream = src->tf_bwd_wstream;
          L.gin = dA[j + 1].p; L.gin_sn = (j + 1 == nrdb) ? 64 * hw : 192 * hw;
          L.dA = dAp.data(); L.cat = catp.data(); L.g_a3 = g_a3.p;
          L.inbox = tf_inbox; L.err = ctx->dev_err_d; L.err_dev = ctx->dev_err_flag;
          L.nrdb = nrdb; L.j0 = jlo; L.j1 = j + 1; L.nimg = std::min(IMGS, N - i0); L.img0 = i0; L.epoch = ++tf_epoch;
          L.rs = rs; L.slope = SLOPE;
          if (src->ev_pack[2]) DBM_HIP(hipStreamWaitEvent(s, src->ev_pack[2], 0));
          ctx->persist_begin(s);
          launch_trunk_fused_bwd(L, s);
          ctx->persist_end(s);
        }
        prev = grp;
        prev_lo = jlo; prev_hi = j + 1;
        final_hi = j + 1;
      }
      // the weight gradients of this dense block (descriptors only; launched with the group)
      const float* Gout = dA[j + 1].p;
      const long gsn = (j + 1 == nrdb) ? 64 * hw : 192 * hw;
      const float sc = (j % 3 == 2) ? rs * rs : rs;
      run_wgrad(layers[L_rdb[j * 5 + 4]], cat[j].p, 192 * hw, h, w, 0, Gout, gsn, h, w, N, sc, &wbs[grp]);
      for (int k = 3; k >= 0; --k)
        run_wgrad(layers[L_rdb[j * 5 + k]], cat[j].p, 192 * hw, h, w, 0, dA[j].p + (long)(64 + 32 * k) * hw, 192 * hw, h, w, N,
                  1.f, &wbs[grp]);
    }
  }
  if (!fused) (owner ? owner : this)->ensure_packed_lazy();
  const int nsplit = fused ? 1 : std::min(trunk_split(N, hw), max_split);
  auto cstream = [&](int c) { return c == 0 ? s : ctx->chain[chain_base + c - 1]; };
  auto chunk = [&](ConvDesc d, int c) {  // descriptor restricted to image range c
    const long n0 = ((long)c * N) / nsplit;
    d.x += n0 * d.xsn; d.y += n0 * d.ysn; d.N = (int)(((long)(c + 1) * N) / nsplit - n0);
    if (d.r1) d.r1 += n0 * d.r1sn;
    if (d.r2) d.r2 += n0 * d.r2sn;
    if (d.mask) d.mask += n0 * d.masksn;
    return d;
  };
  auto join_chains = [&]() { for (int c = 1; c < nsplit; ++c) ctx->fork(cstream(c), s, 7); };
  for (int c = 1; c < nsplit; ++c) ctx->fork(s, cstream(c), 7);
  int prev_grp = fused ? 5 : -1;  // (fused: the last group is still to be launched below)
  for (int j = fused ? -1 : nrdb - 1; j >= 0; --j) {
    // groups of residual-in-residual blocks, shrinking towards the end of the chain: what is still to do once the
    // data-gradient chain has finished (the last group + the pre-residual / input-block batch) is exposed time
    const int r = j / 3;
    const int grp = r == 0 ? 5 : r == 1 ? 4 : r <= 3 ? 3 : (r >= 4 + (n_rrdb - 3) / 2 ? 1 : 2);
    if (prev_grp >= 0 && grp != prev_grp) {
      join_chains();
      ctx->fork_to_side(prev_grp);
      wbs[prev_grp].launch(ctx->side);
    }
    prev_grp = grp;
    const float* Gout = dA[j + 1].p;
    const long gsn = (j + 1 == nrdb) ? 64 * hw : 192 * hw;
    const bool third = (j % 3 == 2), first = (j % 3 == 0);
    const float sc = third ? rs * rs : rs;  // d(out)/d(a5), including the RRDB scaling for the third block
    float* D = dA[j].p;
    const float* C = cat[j].p;
    {  // conv_layer5: out = a5*rs + a0
      const IgLayer& L = layers[L_rdb[j * 5 + 4]];
      run_wgrad(L, C, 192 * hw, h, w, 0, Gout, gsn, h, w, N, sc, &wbs[grp]);
      ConvDesc d;
      memset(&d, 0, sizeof(d));
      d.x = Gout; d.xsn = gsn; d.N = N;
      d.y = D; d.ysn = 192 * hw;
      d.s1 = sc; d.s2 = 1.f;
      d.r1 = Gout; d.r1sn = gsn; d.r1_nch = 64; d.r1s = third ? rs : 1.f;
      d.mask = C; d.masksn = 192 * hw; d.mask_c0 = 160;
      for (int c = 0; c < nsplit; ++c) run_dgrad(L, chunk(d, c), h, w, cstream(c));
    }
    for (int k = 3; k >= 0; --k) {  // conv_layer4 .. conv_layer1
      const int lo = 64 + 32 * k;   // channel offset of a_{k+1} = number of input channels of this conv
      const IgLayer& L = layers[L_rdb[j * 5 + k]];
      run_wgrad(L, C, 192 * hw, h, w, 0, D + (long)lo * hw, 192 * hw, h, w, N, 1.f, &wbs[grp]);
      ConvDesc d;
      memset(&d, 0, sizeof(d));
      d.x = D + (long)lo * hw; d.xsn = 192 * hw; d.N = N;
      d.y = D; d.ysn = 192 * hw; d.accumulate = 1; d.s1 = 1.f; d.s2 = 1.f;
      if (k > 0) {
        d.mask = C; d.masksn = 192 * hw; d.mask_c0 = lo - 32;
      } else {
        if (first) {  // the RRDB skip: d loss / d x += d loss / d (RRDB output)
          d.r1 = dA[j + 3].p; d.r1sn = (j + 3 == nrdb) ? 64 * hw : 192 * hw; d.r1_nch = 64; d.r1s = 1.f;
        }
        if (j == 0) {  // a3 = a1 + ...: add g_a3, then through the pre-residual LeakyReLU (:542)
          d.r2 = g_a3.p; d.r2sn = 64 * hw; d.s2 = 1.f;
          d.mask = cat[0].p; d.masksn = 192 * hw; d.mask_c0 = 0;
        }
      }
      for (int c = 0; c < nsplit; ++c) run_dgrad(L, chunk(d, c), h, w, cstream(c));
    }
  }
  join_chains();
  DBM_MARK(s, "G:backward_trunk_chain");
  // ---- pre_residual_conv_layer and the input block ----
  SmallConvDesc small[4];
  int small_i[4], nsmall = 0;
  {
    const IgLayer& L = layers[L_pre];
    // (its inputs -- a0 and the chain's last output -- are final when the chain is: the pre-residual weight gradient rides in the
    //  trunk's last launch (same kernel form) instead of being a 50-us launch + fold of its own in the serial tail behind it)
    run_wgrad(L, a0.p, 128 * hw, h, w, 0, dA[0].p, 192 * hw, h, w, N, 1.f, &wbs[prev_grp >= 0 ? prev_grp : 6]);
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = dA[0].p; d.xsn = 192 * hw; d.N = N;
    d.y = g_a0.p; d.ysn = 128 * hw; d.s1 = 1.f; d.s2 = 1.f;
    run_dgrad(L, d, h, w);
    struct { const float* in; int Cin, Hin, Win, K, stride; } br[4] = {{in_x.p, 1, H, W, 3, 1},
                                                                    {in_w1.p, 1, 10 * H, 10 * W, 30, 10},
                                                                    {in_w2.p, 2, 2 * H, 2 * W, 6, 2},
                                                                    {in_w3.p, 1, H, W, 3, 1}};
    for (int i = 0; i < 4; ++i) {
      SmallConvDesc q;
      memset(&q, 0, sizeof(q));
      q.x = bw_in[i] ? bw_in[i] : br[i].in;
      q.xsn = (long)br[i].Cin * br[i].Hin * br[i].Win; q.Cin = br[i].Cin; q.Hin = br[i].Hin; q.Win = br[i].Win;
      q.Cout = 32; q.OH = h; q.OW = w; q.KH = q.KW = br[i].K; q.stride = br[i].stride; q.pad = 0; q.N = N;
      if (L_in[i] >= 0) {
        const IgLayer& L = layers[L_in[i]];
        run_wgrad(L, (i == 1 ? colW1 : colW2).p, (long)L.CinP * hw, h, w, 0, g_a0.p + (long)i * 32 * hw, 128 * hw, h, w, N,
                  1.f, &wbs[6]);
      } else {
        small[nsmall] = q;
        small_i[nsmall++] = i;
      }
    }
  }
  DBM_MARK(s, "G:backward_input_block");
  // (Round 4, measured on one box: launching the trunk's weight gradients on the side stream right behind the chain -- before
  //  the pre-residual data gradient above instead of after it -- costs 0.7-1.0 ms per iteration, 8.05 -> 8.73 / 9.04 ms: the
  //  1008 long workgroups then hold every CU while the eight short dependent launches of this tail and the discriminator's
  //  eval-mode pass each wait for a free slot.  The order below stays.)
  ctx->fork_to_side(6);
  hipStream_t wgs = inline_wg ? s : ctx->side;
  DBM_MARK(wgs, "G:side_backlog_done");   // (phase marks on the weight-gradient stream: what stood in front of the trunk's launch is done)
  if (prev_grp >= 0 && !(iter_abl & 2)) wbs[prev_grp].launch(wgs);
  DBM_MARK(wgs, "G:trunk_weight_gradients");
  if (inline_wg && !(iter_abl & 4)) wbs[0].launch(wgs);
  wbs[6].launch(wgs);
  for (int k = 0; k < nsmall; ++k) {  // the two single-channel 3x3 branches of the input block
    const int i = small_i[k];
    launch_smallcin_conv_wgrad(small[k], g_a0.p + (long)i * 32 * hw, 128 * hw, G(T_in[i][0]), G(T_in[i][1]), wgs);
  }
  // input block, pre-residual conv and the trunk group launched last (layer-wise trunk path: the whole trunk)
  if (ctx->comm_in_step) ctx->comm_bucket(grads, rdb_off(final_hi), ctx->side);
  ctx->join_side();  // the optimizer (and the next cleargrads) must see every gradient
}
