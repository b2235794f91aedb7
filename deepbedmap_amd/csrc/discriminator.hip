// DiscriminatorModel (reference srgan_train.py:591-699): parameter table, forward, backward.
#include "model.h"

static const float SLOPE = 0.2f;
static const int DC_O[10] = {64, 64, 128, 128, 128, 256, 256, 512, 512, 512};
static const int DC_C[10] = {1, 64, 64, 128, 128, 128, 256, 256, 512, 512};
static const int DC_K[10] = {3, 4, 3, 4, 3, 4, 3, 4, 3, 4};
static const int DC_S[10] = {1, 2, 1, 2, 1, 2, 1, 2, 1, 2};

Discriminator::Discriminator(dbm_ctx* c) {
  ctx = c;
  type = 1;
  // conv_layer0 has a bias, conv_layer1..9 are nobias=True (:617-634)
  for (int i = 0; i < 10; ++i) {
    const std::string name = "conv_layer" + std::to_string(i);
    add_tensor(name + "/W", {DC_O[i], DC_C[i], DC_K[i], DC_K[i]}, DBM_KIND_PARAM);
    if (i == 0) add_tensor(name + "/b", {DC_O[i]}, DBM_KIND_PARAM);
  }
  T_c0W = tid("conv_layer0/W");
  T_c0b = tid("conv_layer0/b");
  for (int i = 1; i < 10; ++i) {  // :636-644
    const std::string name = "batch_norm" + std::to_string(i);
    T_bn[i][0] = add_tensor(name + "/gamma", {DC_O[i]}, DBM_KIND_PARAM);
    T_bn[i][1] = add_tensor(name + "/beta", {DC_O[i]}, DBM_KIND_PARAM);
  }
  T_l1W = add_tensor("linear_1/W", {100, 512}, DBM_KIND_PARAM);  // :646
  T_l1b = add_tensor("linear_1/b", {100}, DBM_KIND_PARAM);
  T_l2W = add_tensor("linear_2/W", {1, 100}, DBM_KIND_PARAM);  // :647
  T_l2b = add_tensor("linear_2/b", {1}, DBM_KIND_PARAM);
  for (int i = 1; i < 10; ++i) {
    const std::string name = "batch_norm" + std::to_string(i);
    T_bn[i][2] = add_tensor(name + "/avg_mean", {DC_O[i]}, DBM_KIND_PERSISTENT);
    T_bn[i][3] = add_tensor(name + "/avg_var", {DC_O[i]}, DBM_KIND_PERSISTENT);
    T_bn[i][4] = add_tensor(name + "/N", {}, DBM_KIND_PERSISTENT);
  }
  for (int i = 1; i < 10; ++i)
    L_conv[i] = add_iglayer("conv_layer" + std::to_string(i), DC_O[i], DC_C[i], DC_K[i], DC_S[i], 1, false);
  alloc_arenas();
  // Chainer initial state: gamma = 1, avg_var = 1 (beta, avg_mean, N = 0)
  for (int i = 1; i < 10; ++i) {
    launch_fill(P(T_bn[i][0]), DC_O[i], 1.f, ctx->stream);
    launch_fill(S(T_bn[i][3]), DC_O[i], 1.f, ctx->stream);
  }
}

static void layer_dims(int H, int W, int hs[11], int ws[11]) {
  hs[0] = H; ws[0] = W;  // conv0 keeps the size
  hs[1] = H; ws[1] = W;  // h0
  for (int i = 1; i < 10; ++i) {
    hs[i + 1] = (hs[i] + 2 - DC_K[i]) / DC_S[i] + 1;
    ws[i + 1] = (ws[i] + 2 - DC_K[i]) / DC_S[i] + 1;
  }
}

// Eval-mode coefficients of cache slot `slot` (scale = gamma / sqrt(avg_var + eps), shift = beta - avg_mean * scale for the nine
// BatchNorm layers): ONE launch.  forward(bn_train = false) calls it itself unless the caller has done so (coef_ready) -- the deferred
// eval-mode pass of dbm_train_iteration takes its coefficients right behind the discriminator's update, while the running statistics
// and the parameters are still those the reference's call (srgan_train.py:1228) sees, and runs its convolutions later.
void Discriminator::prepare_eval_coeffs(int slot, hipStream_t s) {
  BnEvalJobs ej;
  memset(&ej, 0, sizeof(ej));
  ej.n = 9;
  for (int i = 1; i < 10; ++i) {
    ej.start[i - 1] = ej.total;
    ej.gamma[i - 1] = P(T_bn[i][0]); ej.beta[i - 1] = P(T_bn[i][1]);
    ej.avg_mean[i - 1] = S(T_bn[i][2]); ej.avg_var[i - 1] = S(T_bn[i][3]);
    ej.total += DC_O[i];
  }
  ej.start[9] = ej.total;
  DevBuf& coef = bn_coef[slot];
  coef.ensure(2 * (size_t)ej.total);
  ej.scale = coef.p; ej.shift = coef.p + ej.total;
  launch_bn_eval_coeffs(ej, 1e-5f, s);
}

void Discriminator::forward(int N, int H, int W, const float* img, float* logits, bool bn_train, bool keep, int slot, bool coef_ready) {
  DBM_CHECK(slot >= 0 && slot <= 2 && (slot < 2 || !bn_train), "discriminator cache slot must be 0 or 1 (2: the library's own deferred eval-mode pass)");
  int hs[11], ws[11];
  layer_dims(H, W, hs, ws);  // hs[i+1] = spatial size of h_i
  DBM_CHECK(hs[10] == 1 && ws[10] == 1, "discriminator input must reduce to 1x1 (linear_1 expects 512 features)");
  ensure_packed();
  hipStream_t s = ctx->stream;
  Cache& c = cache[slot];
  const size_t n = (size_t)N;
  c.h[0].ensure(n * 64 * H * W);
  {  // conv_layer0 + LeakyReLU  (:658-659)
    SmallConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = img; d.xsn = (long)H * W; d.Cin = 1; d.Hin = H; d.Win = W;
    d.w = P(T_c0W); d.bias = P(T_c0b);
    d.y = c.h[0].p; d.ysn = 64L * H * W; d.Cout = 64; d.OH = H; d.OW = W;
    d.KH = d.KW = 3; d.stride = 1; d.pad = 1; d.N = N; d.act = 1; d.slope = SLOPE;
    launch_smallcin_conv_fwd(d, s);
  }
  // Eval mode (chainer.config.train = False: the G-step's detached pass, :1228, and the dev-set evaluation): BatchNorm is a
  // per-channel affine map of running statistics -- folded into the convolutions' epilogues with the LeakyReLU (one
  // coefficient launch for the nine layers, no BatchNorm launch, no pre-normalisation planes).
  BnEvalJobs ej;
  memset(&ej, 0, sizeof(ej));
  if (!bn_train) {
    if (!coef_ready) prepare_eval_coeffs(slot, s);
    for (int i = 1; i < 10; ++i) {
      ej.start[i - 1] = ej.total;
      ej.total += DC_O[i];
    }
    ej.start[9] = ej.total;
    DBM_CHECK(bn_coef[slot].n >= 2 * (size_t)ej.total, "eval-mode coefficients of this cache slot were never prepared");
    ej.scale = bn_coef[slot].p; ej.shift = bn_coef[slot].p + ej.total;
  }
  // (libdbm_measure.so only, results wrong: what the deep end -- conv_layer5..9, their BatchNorm launches, the linear layers -- costs INSIDE
  //  the iteration, i.e. the upper bound of what fusing it can bring.  1: training-mode forwards, 2: backward passes, 8: eval-mode forward)
  static const int d_abl = DBM_MEASURE_ENV("D_ABL");
  const bool skip_deep = (d_abl & (bn_train ? 1 : 8)) != 0;
  for (int i = 1; i < 10; ++i) {  // conv -> BatchNorm -> LeakyReLU  (:663-689)
    const IgLayer& L = layers[L_conv[i]];
    if (skip_deep && i >= 5) { c.h[i].ensure((size_t)N * DC_O[i] * hs[i + 1] * ws[i + 1]); c.z[i].ensure((size_t)N * DC_O[i] * hs[i + 1] * ws[i + 1]);
                               c.mean[i].ensure(DC_O[i]); c.istd[i].ensure(DC_O[i]); continue; }
    const int hin = hs[i], win = ws[i], ho = hs[i + 1], wo = ws[i + 1];
    const size_t cnt = n * DC_O[i] * ho * wo;
    c.h[i].ensure(cnt);
    if (!bn_train) {
      ConvDesc d = fwd_desc(L, c.h[i - 1].p, (long)DC_C[i] * hin * win, hin, win, 0, c.h[i].p, (long)DC_O[i] * ho * wo, N);
      d.ch_scale = ej.scale + ej.start[i - 1];
      d.bias = ej.shift + ej.start[i - 1];
      d.act = 1; d.slope = SLOPE;
      launch_igemm_conv(d, s);
      continue;
    }
    c.z[i].ensure(cnt);
    c.mean[i].ensure(DC_O[i]);
    c.istd[i].ensure(DC_O[i]);
    ConvDesc d = fwd_desc(L, c.h[i - 1].p, (long)DC_C[i] * hin * win, hin, win, 0, c.z[i].p, (long)DC_O[i] * ho * wo, N);
    launch_igemm_conv(d, s);
    if (ctx->sync_stats()) {  // statistics of the global batch: local sums -> all-reduce -> apply
      ctx->sync_buf.ensure(3 * 512 + 4);
      launch_bn_sync_stats(c.z[i].p, ctx->sync_buf.p, N, DC_O[i], ho * wo, s);
      ctx->allreduce(ctx->sync_buf.p, 3 * DC_O[i]);
      launch_bn_sync_fwd_apply(c.z[i].p, c.h[i].p, P(T_bn[i][0]), P(T_bn[i][1]), ctx->sync_buf.p, c.mean[i].p, c.istd[i].p,
                               S(T_bn[i][2]), S(T_bn[i][3]), N, DC_O[i], ho * wo, ctx->sync_world, 1e-5f, 0.9f, SLOPE, s, ctx->dev_err_flag);
    } else
      launch_bn_train_fwd(c.z[i].p, c.h[i].p, P(T_bn[i][0]), P(T_bn[i][1]), c.mean[i].p, c.istd[i].p, S(T_bn[i][2]),
                          S(T_bn[i][3]), N, DC_O[i], ho * wo, 1e-5f, 0.9f, SLOPE, s, ctx->dev_err_flag);
  }
  if ((c.N != N || c.H != H || c.W != W) && slot < 2) {
    for (auto& b : wb[slot]) b.reset();
    for (auto& b : wbm) b.reset();
  }  // buffers may move: re-plan the batched weight gradients
  c.l1.ensure(n * 100);
  if (!skip_deep) {
  // linear_1 -> LeakyReLU -> linear_2 (:693-696) as ONE launch (round 6: bitwise the two linear_fwd launches)
  // (DBM_DISC_HEAD_FUSED=0, libdbm_measure.so only: the two-launch form -- A/B)
  static const int head_fused = DBM_TUNE_GETENV("DISC_HEAD_FUSED") ? atoi(DBM_TUNE_GETENV("DISC_HEAD_FUSED")) : 1;
  if (head_fused) {
    launch_disc_head_fwd(c.h[9].p, P(T_l1W), P(T_l1b), P(T_l2W), P(T_l2b), c.l1.p, logits, N, 512, 100, SLOPE, s);
  } else {
    launch_linear_fwd(c.h[9].p, P(T_l1W), P(T_l1b), c.l1.p, N, 512, 100, 1, SLOPE, s);  // :693-695
    launch_linear_fwd(c.l1.p, P(T_l2W), P(T_l2b), logits, N, 100, 1, 0, SLOPE, s);       // :696
  }
  }
  c.N = N; c.H = H; c.W = W;
  c.valid = keep && bn_train;
  if (c.valid) {  // conv_layer0's weight gradient needs the input image: keep a private copy
    if (borrow_images) {   // (the fused steps: the backward pass runs inside the same call, the images outlive it -- no copy launch)
      c.img_src = img;
    } else {
      c.img.ensure(n * H * W);
      DBM_HIP(hipMemcpyAsync(c.img.p, img, n * H * W * sizeof(float), hipMemcpyDeviceToDevice, s));
      c.img_src = c.img.p;
    }
  }
}

// Weight gradients of one layer group go to the side stream once their inputs are final.  Merged mode (the fused
// D-step): the pass that is enqueued first only records an event per group; the pass enqueued second
// (`merge_launcher`) launches the group for both graphs behind both events.
void Discriminator::launch_group(int slot, int g) {
  if (!merge_slots) {
    ctx->fork_to_side(2 + slot);
    wb[slot][g].launch(ctx->side);
    return;
  }
  if (!ev_grp[slot][g]) DBM_HIP(hipEventCreateWithFlags(&ev_grp[slot][g], hipEventDisableTiming));
  DBM_HIP(hipEventRecord(ev_grp[slot][g], ctx->stream));
  if (slot == merge_launcher) {  // the pass that is enqueued second: both events of this step exist now
    DBM_HIP(hipStreamWaitEvent(ctx->side, ev_grp[slot][g], 0));
    if (ev_grp[1 - slot][g]) DBM_HIP(hipStreamWaitEvent(ctx->side, ev_grp[1 - slot][g], 0));
    wbm[g].launch(ctx->side);
    // Data-parallel run: group 0 = conv_layer6..9 is 89 % of the discriminator's parameters, one contiguous range of the
    // gradient arena, and final as soon as this launch is: its all-reduce runs underneath the rest of the backward pass.
    if (g == 0 && ctx->comm_in_step) {
      const size_t lo = tensors[layers[L_conv[6]].wi].off, hi = tensors[layers[L_conv[9]].wi].off + tensors[layers[L_conv[9]].wi].n;
      ctx->comm_bucket(grads + lo, hi - lo, ctx->side);
      comm_sent_lo = lo; comm_sent_hi = hi;
    }
  }
}

static inline int wgroup(int layer) { return layer >= 6 ? 0 : layer >= 4 ? 1 : layer >= 2 ? 2 : 3; }

void Discriminator::backward(int slot, const float* glogits, bool join) {
  Cache& c = cache[slot];
  DBM_CHECK(c.valid, "discriminator backward without a retained training-mode forward");
  mark_grads_touched();
  hipStream_t s = ctx->stream;
  ensure_packed_bwd();   // (callers that run the two graphs' passes on two streams have done this before their fork)
  const int N = c.N;
  int hs[11], ws[11];
  layer_dims(c.H, c.W, hs, ws);
  const size_t n = (size_t)N;
  g_l1[slot].ensure(n * 100);
  g_h[slot][0].ensure(n * 64 * c.H * c.W);
  g_h[slot][1].ensure(n * 64 * c.H * c.W);
  for (int i = 1; i < 10; ++i) g_z[slot][i].ensure(n * DC_O[i] * hs[i + 1] * ws[i + 1]);
  // linear_2, then linear_1 (through its LeakyReLU)
  static const int d_abl = DBM_MEASURE_ENV("D_ABL");   // (libdbm_measure.so only: see forward())
  const bool skip_deep = (d_abl & 2) != 0;
  float* gh = g_h[slot][0].p;
  float* gh_next = g_h[slot][1].p;
  if (!skip_deep) {
  static const int head_fused = DBM_TUNE_GETENV("DISC_HEAD_FUSED") ? atoi(DBM_TUNE_GETENV("DISC_HEAD_FUSED")) : 1;
  if (head_fused && disc_head_bwd_fused_ok(N, 100)) {  // both linear layers' backward as ONE launch (round 6: bitwise the two linear_bwd launches)
    launch_disc_head_bwd(c.h[9].p, P(T_l1W), P(T_l2W), glogits, c.l1.p, gh, G(T_l1W), G(T_l1b), G(T_l2W), G(T_l2b), N, 512, 100, SLOPE, s);
  } else {
    launch_linear_bwd(c.l1.p, P(T_l2W), glogits, nullptr, g_l1[slot].p, G(T_l2W), G(T_l2b), N, 100, 1, SLOPE, s);
    launch_linear_bwd(c.h[9].p, P(T_l1W), g_l1[slot].p, c.l1.p, gh, G(T_l1W), G(T_l1b), N, 512, 100, SLOPE, s);
  }
  }
  for (int i = 9; i >= 1; --i) {
    const IgLayer& L = layers[L_conv[i]];
    const int hin = hs[i], win = ws[i], ho = hs[i + 1], wo = ws[i + 1];
    if (skip_deep && i >= 5) {   // (the layers' weight gradients stay: they run on the side stream and are not what a fused deep end replaces)
      run_wgrad(L, c.h[i - 1].p, (long)DC_C[i] * hin * win, hin, win, 0, g_z[slot][i].p, (long)DC_O[i] * ho * wo, ho, wo, N, 1.f,
                merge_slots ? &wbm[wgroup(i)] : &wb[slot][wgroup(i)]);
      if (wgroup(i - 1) != wgroup(i)) launch_group(slot, wgroup(i));
      continue;
    }
    if (ctx->sync_stats()) {
      ctx->sync_buf.ensure(3 * 512 + 4);
      launch_bn_sync_bwd_sums(c.z[i].p, gh, P(T_bn[i][0]), P(T_bn[i][1]), c.mean[i].p, c.istd[i].p, ctx->sync_buf.p,
                              G(T_bn[i][0]), G(T_bn[i][1]), N, DC_O[i], ho * wo, SLOPE, s);
      ctx->allreduce(ctx->sync_buf.p, 2 * DC_O[i]);
      launch_bn_sync_bwd_apply(c.z[i].p, gh, P(T_bn[i][0]), P(T_bn[i][1]), c.mean[i].p, c.istd[i].p, ctx->sync_buf.p,
                               g_z[slot][i].p, N, DC_O[i], ho * wo, ctx->sync_world, SLOPE, s);
    } else
      launch_bn_train_bwd(c.z[i].p, gh, P(T_bn[i][0]), P(T_bn[i][1]), c.mean[i].p, c.istd[i].p, g_z[slot][i].p, G(T_bn[i][0]),
                          G(T_bn[i][1]), nullptr, N, DC_O[i], ho * wo, SLOPE, s);
    run_wgrad(L, c.h[i - 1].p, (long)DC_C[i] * hin * win, hin, win, 0, g_z[slot][i].p, (long)DC_O[i] * ho * wo, ho, wo, N, 1.f,
              merge_slots ? &wbm[wgroup(i)] : &wb[slot][wgroup(i)]);
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.x = g_z[slot][i].p; d.xsn = (long)DC_O[i] * ho * wo; d.N = N;
    d.y = gh_next; d.ysn = (long)DC_C[i] * hin * win; d.s1 = 1.f; d.s2 = 1.f;
    if (i == 1) {  // through conv_layer0's LeakyReLU
      d.mask = c.h[0].p; d.masksn = 64L * c.H * c.W; d.mask_c0 = 0;
    }
    // this layer closes its group: the group's weight gradients only need the g_z slabs written so far -- they go out
    // BEFORE this layer's data gradient (conv_layer1's, the largest of the chain, used to stand between the last slab and
    // the last group's launch: that much shorter is the wait for the side stream at the end of the step)
    if (i == 1 || wgroup(i - 1) != wgroup(i)) launch_group(slot, wgroup(i));
    run_dgrad(L, d, hin, win);
    float* t = gh; gh = gh_next; gh_next = t;
  }
  {  // conv_layer0 weight / bias gradient
    SmallConvDesc q;
    memset(&q, 0, sizeof(q));
    q.x = c.img_src; q.xsn = (long)c.H * c.W; q.Cin = 1; q.Hin = c.H; q.Win = c.W;
    q.Cout = 64; q.OH = c.H; q.OW = c.W; q.KH = q.KW = 3; q.stride = 1; q.pad = 1; q.N = N;
    c0_scratch[slot].ensure(smallcin_wgrad_scratch_floats(64));
    launch_smallcin_conv_wgrad(q, gh, 64L * c.H * c.W, G(T_c0W), G(T_c0b), s, c0_scratch[slot].p);
  }
  // (conv_layer1..9 weight gradients: one launch per kernel form and layer group, on the side stream -- see the loop)
  if (borrow_images) { c.valid = false; c.img_src = nullptr; }   // (a borrowed image is good for ONE backward pass, inside the borrowing call)
  if (join) ctx->join_side();
}
