// Launchers of the non-GEMM kernels (misc.hip, norm_loss.hip).
#pragma once
#include <cstdlib>
#include "dbm_internal.h"

struct SmallConvDesc {
  const float* x; long xsn; int Cin, Hin, Win;
  const float* w;     // canonical OIHW
  const float* bias;  // may be null
  float* y; long ysn; int Cout, OH, OW;
  int KH, KW, stride, pad;
  int N;
  int act; float slope;
};
void launch_smallcin_conv_fwd(const SmallConvDesc& d, hipStream_t s);
// scratch (optional, smallcin_wgrad_scratch_floats(Cout) floats, private to the launch): the read-dy-once form for <= 9 taps
void launch_smallcin_conv_wgrad(const SmallConvDesc& d, const float* dy, long dysn, float* gW, float* gb, hipStream_t s,
                                float* scratch = nullptr);
size_t smallcin_wgrad_scratch_floats(int Cout);

// DeepbedmapInputBlock on the training tile (11 x 11 -> 9 x 9) as one launch (input_block.hip)
struct InputBlockLaunch {
  const float *x, *w1, *w2, *w3;   // (N,1,11,11), (N,1,110,110), (N,2,22,22), (N,1,11,11), contiguous
  const float *wx, *bx;            // conv_on_X: OIHW (32,1,3,3), bias
  const float *wf1, *b1;           // conv_on_W1: packed forward image [900 (+pad)][32] (IgLayer::wf), bias
  const float *wf2, *b2;           // conv_on_W2: [72 (+pad)][32]
  const float *w3w, *b3;           // conv_on_W3: OIHW
  float* y; long ysn;              // the 128-channel concat (N, 128, plane)
  float* yt;                       // rows form only: the same concat channels-last (N * plane, 128) INSTEAD of y (null: y)
  int N;
};
bool input_block_fused_ok(int H, int W);
void launch_input_block_fused(const InputBlockLaunch& a, hipStream_t s);
// ... on large planes (H x W inputs, the sweep's crops): 32 positions of one output row per workgroup, no im2col image
bool input_block_rows_ok(int H, int W);
void launch_input_block_rows(const InputBlockLaunch& a, int H, int W, hipStream_t s);
void launch_im2col(const float* x, float* col, int N, int Cin, int Hin, int Win, int KH, int KW, int stride, int OH, int OW,
                   int KP, hipStream_t s);
void launch_deform_sample(const float* x, const float* off, float* col, int N, int C, int H, int W, long offsn, hipStream_t s);
void launch_deform_backward(const float* x, const float* off, const float* gcol, const float* w1o, const float* gy,
                            float* gx, float* goff, int N, int C, int H, int W, long offsn, hipStream_t s, hipStream_t aux = nullptr, hipEvent_t* ev = nullptr);
// sampler fused into the GEMM (deform_fused.hip): no column matrix.  C == 64, O == 64 (w = packed [576][64] image) or 1 (w = OIHW).
// xt = the layer input channels-last (N * H * W, 64); yt (optional, O == 64) = the output channels-last as well;
// colout (optional, O == 64) = the sample matrix (N, 576, H, W) as a by-product (a retained pass: the weight gradient reads it)
size_t deform_x3_packed_elems();   // bf16 elements of the split-bf16 weight image of a 64 -> 64 deformable layer
void launch_pack_deform_x3(const float* w_oihw, void* dst, hipStream_t s);
void launch_deform_conv64_x3(const float* xt, const float* off, const void* wx, const float* bias, float* y, float* yt, int N, int H, int W,
                             long offsn, int act, float slope, hipStream_t s, int window = -1);
bool deform_conv_fused_ok(int C, int O);
void launch_nchw_to_nhwc64(const float* x, float* xt, int N, int plane, hipStream_t s);
void launch_deform_conv_fused(const float* xt, const float* off, const float* w, const float* bias, float* y, float* yt, float* colout,
                              int N, int C, int H, int W, long offsn, int O, int act, float slope, hipStream_t s, float* z = nullptr);
// weight gradient of the 64 -> 64 deformable layer with the sampler fused in (no sample matrix): gw (64, 64, 3, 3) / gb (64 or null) += ...
size_t deform_wgrad64_partial_floats(int N, int H, int W);
void launch_deform_wgrad64_fused(const float* xt, const float* off, const float* gy, float* gw, float* gb, float* partial, int N, int H, int W,
                                 long offsn, hipStream_t s);
void launch_deform_bwd64_fused(const float* xt, const float* off, const float* wb, const float* gy, float* gcol, float* goff, int N, int H,
                               int W, long offsn, hipStream_t s);
size_t deform_bwd1_partial_floats(int N, int H, int W);
// the 64 -> 1 layer's backward in the premultiplied form (round 5): z = the forward's premultiplied planes (N, 9, plane), Gt = scratch of
// N * 9 * plane floats, csr_ws = deform_csr_workspace_floats floats; goff and gx (N, 64, plane) overwritten, gw / gb accumulated
void launch_deform_bwd1_premul(const float* xt, const float* off, const float* w, const float* gy, const float* z, float* goff, float* gx,
                               float* gw, float* gb, float* partial, float* csr_ws, float* Gt, int N, int H, int W, long offsn,
                               hipStream_t s, bool lists_built = false);
void launch_deform1_premul(const float* xt, const float* w, float* z, int N, int H, int W, int O, hipStream_t s);
void launch_deform_csr_gather1(const float* off, const float* gy, float* G, int N, int H, int W, long offsn, hipStream_t s, float* ws,
                               bool lists_built = false);
// the sampling lists alone (what launch_deform_csr_gather1 / launch_deform_input_grad build first unless told `lists_built`): they depend on
// the offsets only, so a retained forward can have them built beside its own tail instead of inside the backward pass
bool deform_csr_lists_ok(int C, int H, int W);
void launch_deform_csr_build(const float* off, float* ws, int N, int H, int W, long offsn, hipStream_t s);
void launch_deform_bwd1_fused(const float* xt, const float* off, const float* w, const float* gy, float* goff, float* gw, float* gb,
                              float* partial, int N, int H, int W, long offsn, hipStream_t s);
bool deform_input_grad_ok(int C, int H, int W);
// ws (optional, deform_csr_workspace_floats floats): the sampling lists are built once per (image, tap) there and a
// register-only kernel gathers (otherwise every channel-group workgroup rebuilds them in LDS)
size_t deform_csr_workspace_floats(int N, int H, int W);
void launch_deform_input_grad(const float* x, const float* off, const float* gcol, const float* w1o, const float* gy, float* gx, int N,
                              int C, int H, int W, long offsn, hipStream_t s, float* ws = nullptr, bool lists_built = false);
void launch_gemv_cols(const float* col, const float* w, const float* bias, float* y, int N, int K, int plane, hipStream_t s);
void launch_gemv_cols_wgrad(const float* col, const float* gy, float* gw, float* gb, int N, int K, int plane, hipStream_t s);
void launch_sumpool2(const float* g, const float* mask, float* out, long nc, int H, int W, float slope, hipStream_t s);

// ---- norm_loss.hip ----
// BatchNorm (training): per-channel batch statistics of z [N,C,plane]; writes mean/inv_std, updates running stats,
// then y = lrelu(gamma*(z-mean)*inv_std + beta).
void launch_bn_train_fwd(const float* z, float* y, const float* gamma, const float* beta, float* mean, float* inv_std,
                         float* avg_mean, float* avg_var, int N, int C, int plane, float eps, float decay, float slope,
                         hipStream_t s, const int* hold = nullptr);  // hold: device word; non-zero = no running-average update
struct BnEvalJobs {   // every BatchNorm layer of a model: layer l covers elements [start[l], start[l + 1]) of scale / shift
  static const int MAXL = 12;
  int n, total;
  int start[MAXL + 1];
  const float* gamma[MAXL]; const float* beta[MAXL]; const float* avg_mean[MAXL]; const float* avg_var[MAXL];
  float* scale; float* shift;
};
void launch_bn_eval_coeffs(const BnEvalJobs& jobs, float eps, hipStream_t s);
// backward through lrelu + BN(train): gz = d loss/d z ; ggamma/gbeta accumulated (+=)
void launch_bn_train_bwd(const float* z, const float* gh, const float* gamma, const float* beta, const float* mean,
                         const float* inv_std, float* gz, float* ggamma, float* gbeta, float* scratch, int N, int C,
                         int plane, float slope, hipStream_t s);
// Linear layers of the discriminator head (tiny): y[n][o] = act(b[o] + sum_k W[o][k] x[n][k])
void launch_linear_fwd(const float* x, const float* W, const float* b, float* y, int N, int K, int O, int act,
                       float slope, hipStream_t s);
// gx[n][k] = sum_o gyz[n][o] W[o][k]; gW[o][k] += sum_n gyz[n][o] x[n][k]; gb[o] += sum_n gyz[n][o]
// where gyz = gy * lrelu'(y) if y_act != null else gy
void launch_linear_bwd(const float* x, const float* W, const float* gy, const float* y_act, float* gx, float* gW,
                       float* gb, int N, int K, int O, float slope, hipStream_t s);
// the discriminator's head (linear_1 -> LeakyReLU -> linear_2) as one launch per pass; bitwise the two-launch results
void launch_disc_head_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, float* l1, float* logits,
                          int N, int K, int O, float slope, hipStream_t s);
bool disc_head_bwd_fused_ok(int N, int O);
void launch_disc_head_bwd(const float* x, const float* W1, const float* W2, const float* glogits, const float* l1, float* gx, float* gW1,
                          float* gb1, float* gW2, float* gb2, int N, int K, int O, float slope, hipStream_t s);

// RaGAN discriminator loss (srgan_train.py:960-1009) on N real + N fake logits.
// out[0] = loss, out[1] = binary accuracy (train_eval_discriminator :1156-1158); g_real/g_fake may be null.
// real_targets / fake_targets (device int32[N], may be null): per-sample targets (-1 = ignored) instead of the constants.
void launch_ragan_loss(const float* real, const float* fake, int N, int real_target, int fake_target, float* out,
                       float* g_real, float* g_fake, hipStream_t s, const int* real_targets = nullptr,
                       const int* fake_targets = nullptr);
// mean SSIM's per-image sums (sums[4 n + 2]) for any window_size <= 64 / stride (metric only; the loss kernel is 9 / 1)
void launch_ssim_general(const float* y, const float* t, int N, int H, int W, int ws, int stride, int uniform, float* sums,
                         hipStream_t s);

// Generator loss terms on y_pred vs y_true [N,1,H,W] and x_topo = X[:, :, 1:-1, 1:-1] (srgan_train.py:841-902).
// sums[0..4] = sum|y-t|, sum|pool4(y)-x|, sum ssim_map, sum (y-t)^2, (unused); gy (may be null) receives
// cw*dL1 + tw*dTopo - sw*dSSIM.  X is the full [N,1,H/4+2,W/4+2] input tile.
void launch_gen_loss(const float* y, const float* t, const float* X, int N, int H, int W, float cw, float tw, float sw,
                     const float* win1d, float* sums, float* gy, hipStream_t s);

// Adam (Chainer form, srgan_train.py:1043-1048): one fused pass over the flat arenas.
// skip (may be null): device word; while it is non-zero the update is a no-op (a persistent kernel gave up: the
// gradients of this iteration are invalid, parameters and moments must not be touched).  It is sampled once per launch
// into skipped[1] (adam_gate_kernel): an update is all or nothing.
void launch_adam(float* p, const float* g, float* m, float* v, long n, float alpha_t, float one_minus_beta1,
                 float one_minus_beta2, float eps, float gscale, hipStream_t s, const int* skip = nullptr,
                 int* skipped = nullptr);  // skipped[0] counts the no-op launches, skipped[1] is the launch's gate word
void launch_fill(float* p, long n, float v, hipStream_t s);
void launch_clip_min(float* p, long n, float lo, hipStream_t s);  // p = max(p, lo) in place, NaN kept (np.clip)
void launch_gather_rows(const void* src, void* dst, const int* d_idx, int n, size_t row_bytes, hipStream_t s);
int sqdiff_blocks(long n);  // partial sums launch_sqdiff writes to out[0..blocks)
void launch_sqdiff(const float* a, const float* b, long n, float* out, hipStream_t s);

// ---- fused RRDB trunk forward on 9x9 planes (trunk_fused.hip) ----
#define TRUNK_FUSED_MAXCAT 64
struct TrunkFusedLaunch {
  const float* wstream;  // trunk_fused_stream_floats(nrdb) floats, written by launch_pack_trunk_fused
  const float* bstream;  // nrdb * 192 floats
  const float* in;       // (N, 192, 81) concat buffer whose channels 0..63 hold the trunk input
  float* const* cat;     // HOST table of nrdb + 1 concat buffers (training: every layer output is kept), or null
  float* out;            // cat == null: concat buffer receiving the trunk output in channels 0..63
  unsigned long long* inbox;  // trunk_fused_inbox_bytes(images per launch)
  int* err;              // host-mapped word raised when a neighbour never answered (bounded spins)
  int* err_dev;          // the same flag in device memory: the optimizer kernels skip their update while it is set
  int nrdb, nimg, img0, epoch;
  float rs, slope;
  int no_helper = 0;     // 1: never the four-workgroups-per-image form (data-parallel runs: RCCL's kernels need compute units too)
};
size_t trunk_fused_stream_floats(int nrdb);
size_t trunk_fused_inbox_bytes(int nimg);
void launch_pack_trunk_fused(const float* const* d_wsrc, const float* const* d_bsrc, float* wstream, float* bstream, int nrdb,
                             hipStream_t s);
void launch_trunk_fused(const TrunkFusedLaunch& L, hipStream_t s);

// ---- fused data-gradient chain of the RRDB trunk on 9x9 planes (trunk_fused_bwd.hip) ----
struct TrunkFusedBwdLaunch {
  const float* wstream;      // trunk_fused_bwd_stream_floats(nrdb), written by launch_pack_trunk_fused_bwd
  const float* gin;          // gradient w.r.t. the output of dense block j1 - 1 (channels 0..63), image stride gin_sn
  long gin_sn;
  float* const* dA;          // HOST table: dA[j] (N, 192, 81), j < nrdb
  const float* const* cat;   // HOST table: forward concat buffers
  const float* g_a3;         // (N, 64, 81): added to the trunk input gradient at j == 0
  unsigned long long* inbox;
  int* err;
  int* err_dev;
  int nrdb, j0, j1;          // dense blocks j1 - 1 ... j0 (both multiples of 3)
  int nimg, img0, epoch;
  float rs, slope;
};
size_t trunk_fused_xcc_offset(int nimg_alloc);  // granules in front of the XCC_ID table at the end of an inbox buffer
int trunk_local_stores();                       // DBM_TRUNK_LOCAL_ST (default 1): same-XCD exchange stores stay in the L2
extern bool g_trunk_local_off;                  // ... until a persistent kernel has timed out once in this process
size_t trunk_fused_bwd_stream_floats(int nrdb);
void launch_pack_trunk_fused_bwd(const float* const* d_wsrc, float* wstream, int nrdb, hipStream_t s);
void launch_trunk_fused_bwd(const TrunkFusedBwdLaunch& L, hipStream_t s);

// ---- sync_batch_stats (norm_loss.hip): BatchNorm / RaGAN statistics over the global batch of a data-parallel run ----
void launch_bn_sync_stats(const float* z, float* buf, int N, int C, int plane, hipStream_t s);
void launch_bn_sync_fwd_apply(const float* z, float* y, const float* gamma, const float* beta, const float* buf, float* mean,
                              float* inv_std, float* avg_mean, float* avg_var, int N, int C, int plane, int world, float eps,
                              float decay, float slope, hipStream_t s, const int* hold = nullptr);
void launch_bn_sync_bwd_sums(const float* z, const float* gh, const float* gamma, const float* beta, const float* mean,
                             const float* inv_std, float* buf, float* ggamma, float* gbeta, int N, int C, int plane, float slope,
                             hipStream_t s);
void launch_bn_sync_bwd_apply(const float* z, const float* gh, const float* gamma, const float* beta, const float* mean,
                              const float* inv_std, const float* buf, float* gz, int N, int C, int plane, int world, float slope,
                              hipStream_t s);
void launch_ragan_sync_sums(const float* real, const float* fake, int N, float* buf, hipStream_t s);
void launch_ragan_sync_loss(const float* real, const float* fake, int N, int world, int real_target, int fake_target, float* buf,
                            float* out, hipStream_t s);
void launch_ragan_sync_grad(const float* real, const float* fake, int N, int world, int real_target, int fake_target,
                            const float* buf, float* g_real, float* g_fake, hipStream_t s);

// ---- channels-last bf16 3x3 convolution for the trunk of the bf16 area sweep on large planes (conv_cl16.hip) ----
struct ClConvLaunch {
  const void* x; int xc;     // input NHWC bf16 (xc channels per pixel), channels [0, Cin) are read
  int Cin, Cout;             // Cin % 32 == 0, Cout 32 or 64
  const void* w;             // packed by launch_pack_cl16
  const float* bias;
  void* y16; int yc, y0;     // bf16 output (NHWC, yc channels per pixel, first channel y0) or null
  float* y32;                // fp32 output (NHWC, 64 channels per pixel) or null
  const float* r1; float s1; // v = s1 * (acc + bias) + r1 (NHWC fp32, 64 channels), if r1
  const float* r2; float s2; // v = s2 * v + r2, if r2
  int act; float slope;
  int N, H, W;
  const void* zeros;         // >= 16 bytes of device zeros (dbm_ctx::zeros)
};
size_t cl16_packed_elems(int Cin, int Cout);   // bf16 elements of a layer's packed image
void launch_pack_cl16(const float* w_oihw, void* dst, int O, int C, hipStream_t s);
void launch_conv_cl16(const ClConvLaunch& L, hipStream_t s);
// (N, 64, plane) fp32 [image stride xsn] -> NHWC fp32 (res, may be null) and NHWC bf16 channels 0..63 of a buffer with `ac`
// channels per pixel (act, may be null); and back
void launch_nchw_to_cl(const float* x, long xsn, float* res, void* act, int ac, int N, int plane, hipStream_t s, int nch = 64);
void launch_cl_to_nchw(const float* res, float* y, long ysn, int N, int plane, hipStream_t s, int nch = 64);

// split-bf16 (three bf16 MFMAs per product: 2^-16 operand precision) 3x3 convolution on NHWC fp32 activations: the layers on
// the signal path of the bf16 sweep (upsampling and offset convolutions)
struct ClX3Launch {
  const float* x; int xc;    // input NHWC fp32 (xc channels per pixel), channels [0, Cin) are read
  int Cin, Cout;             // Cin % 16 == 0, Cout <= 64
  int ups;                   // 1: x is the (H / 2, W / 2) plane (nearest x2 folded in)
  const void* w;             // packed by launch_pack_cl16x3
  const float* bias;
  float* y32; int yc;        // NHWC fp32 output (yc channels per pixel; all 32 * ceil(Cout / 32) channels are written) or null
  const float* r1; int r1c;  // optional residual, NHWC fp32 (r1c channels per pixel): v = conv + bias + r1 (before the activation)
  void* y16; int y16c;       // optional second output: the same values as bf16, NHWC with y16c channels per pixel (channels [0, 32 MT))
  float* yp; long ysn; int ypc;  // channel planes yp[n * ysn + co * H * W + pixel], co < ypc, or null
  int act; float slope;
  int N, H, W;               // output plane
};
size_t cl16x3_packed_elems(int Cin, int Cout);
void launch_pack_cl16x3(const float* w_oihw, void* dst, int O, int C, hipStream_t s);
void launch_conv_cl16x3(const ClX3Launch& L, hipStream_t s);
