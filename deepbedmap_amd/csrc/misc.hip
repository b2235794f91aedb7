// Non-GEMM kernels of the hot path: few-input-channel convolutions (input block, D conv0),
// deformable-convolution sampler (forward / backward), 2x2 sum-pool (backward of the nearest
// upsample), small dense layers.  All are HBM/L2-bound VALU kernels: one position per lane,
// coalesced along the innermost (x) axis of the NCHW fp32 tensors.
#include "dbm_internal.h"
#include "deform_geom.h"
#include "kernels.h"
#include <cstdlib>

// ----------------------------------------------------------------------------------------------
// Convolution with 1 or 2 input channels (reference srgan_train.py:223-254 input block, :617-625
// discriminator conv_layer0).  Each thread owns one output position and 8 output channels.
// ----------------------------------------------------------------------------------------------
// K3: 3x3 kernels (the discriminator's conv_layer0, the input block's 3x3 branches) with the tap loops unrolled and branch-free -- every
// input value and every weight of a channel requested together (the general form's run-time loops with their `continue` made nine
// dependent round trips of a 1-channel 3x3 layer: 14.8 us for a 21 MB output).
template <bool K3>
__global__ __launch_bounds__(256) void smallcin_conv_fwd_kernel(const SmallConvDesc d) {
  const int plane = d.OH * d.OW;
  const long P = (long)blockIdx.x * 64 + (threadIdx.x & 63);
  const int cg = threadIdx.x >> 6;  // wave-uniform: 8-channel group within the 32 handled per block
  const int co0 = blockIdx.y * 32 + cg * 8;
  if (P >= (long)d.N * plane || co0 >= d.Cout) return;
  const int n = (int)(P / plane);
  const int r = (int)(P - (long)n * plane);
  const int a = r / d.OW, b = r - a * d.OW;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d.bias ? d.bias[co0 + i] : 0.f;
  const int K = d.Cin * d.KH * d.KW;
  if (K3) {
    for (int c = 0; c < d.Cin; ++c) {
      const float* xc = d.x + (long)n * d.xsn + (long)c * d.Hin * d.Win;
      float xv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = a * d.stride - d.pad + t / 3, ix = b * d.stride - d.pad + t % 3;
        const bool in = (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        const float v = xc[in ? (long)iy * d.Win + ix : 0];
        xv[t] = in ? v : 0.f;
      }
      const float* wr = d.w + (long)co0 * K + c * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t)   // (tap order ky, kx as in the general form: the same sums)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(wr[(long)i * K + t], xv[t], acc[i]);
    }
  } else
  for (int c = 0; c < d.Cin; ++c) {
    const float* xc = d.x + (long)n * d.xsn + (long)c * d.Hin * d.Win;
    for (int ky = 0; ky < d.KH; ++ky) {
      const int iy = a * d.stride - d.pad + ky;
      if ((unsigned)iy >= (unsigned)d.Hin) continue;
      const float* xr = xc + (long)iy * d.Win;
      const float* wr = d.w + (long)co0 * K + (c * d.KH + ky) * d.KW;
      for (int kx = 0; kx < d.KW; ++kx) {
        const int ix = b * d.stride - d.pad + kx;
        const float v = ((unsigned)ix < (unsigned)d.Win) ? xr[ix] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(wr[(long)i * K + kx], v, acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = acc[i];
    if (d.act) v = v >= 0.f ? v : d.slope * v;
    d.y[(long)n * d.ysn + (long)(co0 + i) * plane + r] = v;
  }
}

void launch_smallcin_conv_fwd(const SmallConvDesc& d, hipStream_t s) {
  if (dbm_abl_skip() & 128) return;  // (libdbm_measure.so only)
  DBM_CHECK(d.Cout % 8 == 0, "smallcin conv: Cout must be a multiple of 8");
  const long total = (long)d.N * d.OH * d.OW;
  dim3 grid((unsigned)((total + 63) / 64), (unsigned)((d.Cout + 31) / 32));
  if (d.KH == 3 && d.KW == 3) hipLaunchKernelGGL(smallcin_conv_fwd_kernel<true>, grid, dim3(256), 0, s, d);
  else hipLaunchKernelGGL(smallcin_conv_fwd_kernel<false>, grid, dim3(256), 0, s, d);
  DBM_HIP(hipGetLastError());
}

// gW[o][c][ky][kx] += sum_{n,a,b} dy[n][o][a][b] * x[n][c][a*s-p+ky][b*s-p+kx];  gb[o] += sum dy.
// One WORKGROUP per weight element (and one per bias element): its 256 threads stride over the output positions
// and the partial sums are folded by a fixed shuffle / LDS tree -- no K split across workgroups, so the one atomic
// per element only orders separate launches (the two graphs of the discriminator: a + b == b + a).  Reproducible.
__global__ __launch_bounds__(256) void smallcin_conv_wgrad_kernel(const SmallConvDesc d, const float* __restrict__ dy,
                                                                  long dysn, float* gW, float* gb) {
  __shared__ float sh[4];
  const int K = d.Cin * d.KH * d.KW;
  const int e = blockIdx.x;
  const int plane = d.OH * d.OW;
  const bool bias = e >= d.Cout * K;
  const int o = bias ? e - d.Cout * K : e / K;
  const int k = bias ? 0 : e - o * K;
  const int c = k / (d.KH * d.KW), kr = k - c * d.KH * d.KW;
  const int ky = kr / d.KW, kx = kr - ky * d.KW;
  // lanes run along the flattened (n, a, b) positions: coalesced dy reads; two multiply-high divisions per element
  const unsigned total = (unsigned)d.N * (unsigned)plane;
  const unsigned planeM = 0xffffffffu / (unsigned)plane, owM = 0xffffffffu / (unsigned)d.OW;  // floor((2^32-1)/d): <= 1 short
  float acc = 0.f;
  // (eight positions per thread and trip, all sixteen loads requested before the first multiply-add: the one-position form was twenty
  //  dependent pairs of round trips per thread -- 14.5 us, twice, at the very end of a training iteration.  Same order of sums.)
  for (unsigned P0 = threadIdx.x; P0 < total; P0 += 256 * 8) {
    float g[8], xv[8];
    bool use[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned Pu = P0 + 256u * (unsigned)u;
      const bool in = Pu < total;
      const unsigned P = in ? Pu : 0u;
      unsigned n = __umulhi(P, planeM);
      unsigned r = P - n * (unsigned)plane;
      if (r >= (unsigned)plane) { ++n; r -= (unsigned)plane; }
      g[u] = dy[(long)n * dysn + (long)o * plane + r];
      unsigned a = __umulhi(r, owM);
      unsigned b = r - a * (unsigned)d.OW;
      if (b >= (unsigned)d.OW) { ++a; b -= (unsigned)d.OW; }
      const int iy = (int)a * d.stride - d.pad + ky, ix = (int)b * d.stride - d.pad + kx;
      const bool inside = !bias && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
      xv[u] = d.x[inside ? (long)n * d.xsn + (long)c * d.Hin * d.Win + (long)iy * d.Win + ix : 0];
      use[u] = in && (bias || inside);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float t = bias ? acc + g[u] : fmaf(g[u], xv[u], acc);
      acc = use[u] ? t : acc;
    }
  }
  for (int s2 = 32; s2 > 0; s2 >>= 1) acc += __shfl_down(acc, s2, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (bias) atomicAdd(gb + o, v);
    else atomicAdd(gW + e, v);
  }
}

// Few taps (Cin * KH * KW <= 9: conv_layer0 of the discriminator): a workgroup per (output channel, position slice) keeps
// all nine sums and the bias sum in registers, so dy is read ONCE instead of ten times; the slices' partial sums go to
// a scratch buffer and a second kernel adds them in slice order (reproducible like the kernel above; 249 -> ~30 us).
constexpr int SCW_SLICES = 64;
constexpr int SCW_OG = 8;  // output channels per workgroup of the 3 x 3 form
// K3 = true: the geometry is known at compile time (one input channel, 3 x 3, stride 1, pad 1 = conv_layer0): without it
// every tap decodes (c, ky, kx) with runtime integer divisions -- ~600 VALU instructions per position against ten FMAs.
// The 3 x 3 form also takes SCW_OG output channels per workgroup (blockIdx.x = channel group): a position's nine input
// values are loaded once for all of them (the loads of x, not of dy, were the bulk of the instructions: 49 -> ~15 us).
template <bool K3>
__global__ __launch_bounds__(256) void smallcin_wgrad_partial_kernel(const SmallConvDesc d, const float* __restrict__ dy,
                                                                     long dysn, float* __restrict__ partial) {
  constexpr int OG = K3 ? SCW_OG : 1;
  __shared__ float sh[4][OG * 10];
  const int K = K3 ? 9 : d.Cin * d.KH * d.KW;  // <= 9
  const int o0 = blockIdx.x * OG, z = blockIdx.y;
  const int plane = d.OH * d.OW;
  const unsigned total = (unsigned)d.N * (unsigned)plane;
  const unsigned chunk = (total + SCW_SLICES - 1) / SCW_SLICES;
  const unsigned p0 = z * chunk, p1 = min(total, p0 + chunk);
  const unsigned planeM = 0xffffffffu / (unsigned)plane, owM = 0xffffffffu / (unsigned)d.OW;
  float acc[OG][10];
#pragma unroll
  for (int oo = 0; oo < OG; ++oo)
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[oo][k] = 0.f;
  for (unsigned P = p0 + threadIdx.x; P < p1; P += 256) {
    unsigned n = __umulhi(P, planeM);
    unsigned r = P - n * (unsigned)plane;
    if (r >= (unsigned)plane) { ++n; r -= (unsigned)plane; }
    unsigned a = __umulhi(r, owM);
    unsigned b = r - a * (unsigned)d.OW;
    if (b >= (unsigned)d.OW) { ++a; b -= (unsigned)d.OW; }
    const float* xn = d.x + (long)n * d.xsn;
    const float* dyp = dy + (long)n * dysn + (long)o0 * plane + r;
    if constexpr (K3) {
      const float* xc = xn + (int)a * d.Win + (int)b;
      float xv[9];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = (int)a - 1 + ky, ix = (int)b - 1 + kx;
          xv[ky * 3 + kx] = ((unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win) ? xc[(ky - 1) * d.Win + (kx - 1)] : 0.f;
        }
#pragma unroll
      for (int oo = 0; oo < OG; ++oo) {
        const float g = dyp[(long)oo * plane];
        acc[oo][9] += g;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[oo][k] = fmaf(g, xv[k], acc[oo][k]);
      }
    } else {
      const float g = dyp[0];
      acc[0][9] += g;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (k < K) {
          const int c = k / (d.KH * d.KW), kr = k - c * d.KH * d.KW;
          const int ky = kr / d.KW, kx = kr - ky * d.KW;
          const int iy = (int)a * d.stride - d.pad + ky, ix = (int)b * d.stride - d.pad + kx;
          if ((unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win)
            acc[0][k] = fmaf(g, xn[(long)c * d.Hin * d.Win + (long)iy * d.Win + ix], acc[0][k]);
        }
      }
    }
  }
#pragma unroll
  for (int oo = 0; oo < OG; ++oo)
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      float v = acc[oo][k];
      for (int s2 = 32; s2 > 0; s2 >>= 1) v += __shfl_down(v, s2, 64);
      if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][oo * 10 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < OG * 10) {
    const int k = threadIdx.x;  // (channel oo = k / 10, sum k % 10)
    partial[((long)z * d.Cout + o0) * 10 + k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
  }
}

// one wavefront per (output channel, sum): lane z holds slice z, the shuffle tree adds them in a fixed order
static_assert(SCW_SLICES == 64, "smallcin_wgrad_fold_kernel: one lane per slice");
__global__ __launch_bounds__(640) void smallcin_wgrad_fold_kernel(const float* __restrict__ partial, int Cout, int K, float* gW,
                                                                  float* gb) {
  const int o = blockIdx.x, k = threadIdx.x >> 6, z = threadIdx.x & 63;
  float v = partial[((long)z * Cout + o) * 10 + k];
  for (int s2 = 32; s2 > 0; s2 >>= 1) v += __shfl_down(v, s2, 64);
  if (z != 0) return;
  if (k == 9) { if (gb) atomicAdd(gb + o, v); }
  else if (k < K) atomicAdd(gW + o * K + k, v);
}

size_t smallcin_wgrad_scratch_floats(int Cout) { return (size_t)SCW_SLICES * Cout * 10; }

void launch_smallcin_conv_wgrad(const SmallConvDesc& d, const float* dy, long dysn, float* gW, float* gb,
                                hipStream_t s, float* scratch) {
  if (dbm_abl_skip() & 128) return;  // (libdbm_measure.so only)
  const int K = d.Cin * d.KH * d.KW;
  if (scratch && K <= 9 && (long)d.N * d.OH * d.OW >= 16384) {
    if (d.Cin == 1 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.Cout % SCW_OG == 0)
      hipLaunchKernelGGL(smallcin_wgrad_partial_kernel<true>, dim3(d.Cout / SCW_OG, SCW_SLICES), dim3(256), 0, s, d, dy, dysn, scratch);
    else
      hipLaunchKernelGGL(smallcin_wgrad_partial_kernel<false>, dim3(d.Cout, SCW_SLICES), dim3(256), 0, s, d, dy, dysn, scratch);
    hipLaunchKernelGGL(smallcin_wgrad_fold_kernel, dim3(d.Cout), dim3(640), 0, s, scratch, d.Cout, K, gW, gb);
    DBM_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(smallcin_conv_wgrad_kernel, dim3(d.Cout * K + (gb ? d.Cout : 0)), dim3(256), 0, s, d, dy, dysn, gW, gb);
  DBM_HIP(hipGetLastError());
}

// col[n][k][p] = x[n][c][a*s+ky][b*s+kx] for k = (c*KH+ky)*KW+kx < K, 0 for K <= k < KP (valid convolution, no padding).
// Turns the two wide-kernel input-block branches (k30 s10 on REMA, k6 s2 on MEaSUREs, srgan_train.py:231-246) into
// 900- / 72-deep GEMMs for the MFMA kernels; the OIHW weight flattening is exactly this k order.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int N, int Cin,
                                                     int Hin, int Win, int KH, int KW, int stride, int OH, int OW, int K,
                                                     int KP) {
  const int plane = OH * OW;
  const long total = (long)N * KP * plane;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int p = (int)(e % plane);
    const int k = (int)((e / plane) % KP);
    const int n = (int)(e / ((long)plane * KP));
    float v = 0.f;
    if (k < K) {
      const int c = k / (KH * KW), kr = k - c * KH * KW;
      const int ky = kr / KW, kx = kr - ky * KW;
      const int a = p / OW, b = p - a * OW;
      v = x[((long)n * Cin + c) * Hin * Win + (long)(a * stride + ky) * Win + b * stride + kx];
    }
    col[e] = v;
  }
}

void launch_im2col(const float* x, float* col, int N, int Cin, int Hin, int Win, int KH, int KW, int stride, int OH, int OW,
                   int KP, hipStream_t s) {
  if (dbm_abl_skip() & 256) return;  // (libdbm_measure.so only)
  const long total = (long)N * KP * OH * OW;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, col, N, Cin, Hin, Win, KH, KW, stride, OH, OW,
                     Cin * KH * KW, KP);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// Deformable convolution sampler (reference srgan_train.py:506-523, :572-574; Chainer
// deformable_convolution_2d_sampler + spatial_transformer_sampler semantics).
// ----------------------------------------------------------------------------------------------
// col[n][c*9+t][p] = bilinear sample of x[n][c] at (tap t position + offset)
__global__ __launch_bounds__(256) void deform_sample_kernel(const float* __restrict__ x, const float* __restrict__ off,
                                                            float* __restrict__ col, int N, int C, int H, int W,
                                                            long offsn) {
  const int plane = H * W;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * 9 * plane) return;
  const int p = (int)(e % plane);
  const int t = (int)((e / plane) % 9);
  const int n = (int)(e / (9L * plane));
  const int a = p / W, b = p - a * W;
  const float* on = off + (long)n * offsn;
  const DeformGeom g = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
  const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
  const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
  const float w1 = g.wu1 * g.wv1, w2 = g.wu0 * g.wv1, w3 = g.wu1 * g.wv0, w4 = g.wu0 * g.wv0;
  const float* xn = x + (long)n * C * plane;
  float* cn = col + ((long)n * C * 9 + t) * plane + p;
  for (int c = 0; c < C; ++c) {
    const float* xc = xn + (long)c * plane;
    const float x1 = o1 >= 0 ? xc[o1] : 0.f, x2 = o2 >= 0 ? xc[o2] : 0.f;
    const float x3 = o3 >= 0 ? xc[o3] : 0.f, x4 = o4 >= 0 ? xc[o4] : 0.f;
    cn[(long)c * 9 * plane] = w1 * x1 + w2 * x2 + w3 * x3 + w4 * x4;
  }
}

void launch_deform_sample(const float* x, const float* off, float* col, int N, int C, int H, int W, long offsn,
                          hipStream_t s) {
  const long total = (long)N * 9 * H * W;
  hipLaunchKernelGGL(deform_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, off, col, N, C,
                     H, W, offsn);
  DBM_HIP(hipGetLastError());
}

// Backward of the sampler.  gcol[n][c*9+t][p] is either read (gcol != null) or, for a single
// output channel, formed on the fly as w1o[c*9+t] * gy[n][p].  Scatters into gx (atomics; gx
// must be zero-initialised or hold the gradient it accumulates onto) and writes goff[n][0:18].
__global__ __launch_bounds__(256) void deform_backward_kernel(const float* __restrict__ x, const float* __restrict__ off,
                                                              const float* __restrict__ gcol,
                                                              const float* __restrict__ w1o,
                                                              const float* __restrict__ gy, float* gx,
                                                              float* __restrict__ goff, int N, int C, int H, int W,
                                                              long offsn) {
  const int plane = H * W;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * 9 * plane) return;
  const int p = (int)(e % plane);
  const int t = (int)((e / plane) % 9);
  const int n = (int)(e / (9L * plane));
  const int a = p / W, b = p - a * W;
  const float* on = off + (long)n * offsn;
  const DeformGeom g = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
  const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
  const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
  const float w1 = g.wu1 * g.wv1, w2 = g.wu0 * g.wv1, w3 = g.wu1 * g.wv0, w4 = g.wu0 * g.wv0;
  const float* xn = x + (long)n * C * plane;
  float* gxn = gx + (long)n * C * plane;
  const float gyv = gy ? gy[(long)n * plane + p] : 0.f;
  float gu = 0.f, gv = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* xc = xn + (long)c * plane;
    float* gxc = gxn + (long)c * plane;
    const float gq = gcol ? gcol[((long)n * C * 9 + (long)c * 9 + t) * plane + p] : w1o[c * 9 + t] * gyv;
    const float x1 = o1 >= 0 ? xc[o1] : 0.f, x2 = o2 >= 0 ? xc[o2] : 0.f;
    const float x3 = o3 >= 0 ? xc[o3] : 0.f, x4 = o4 >= 0 ? xc[o4] : 0.f;
    gu += gq * (-g.wv1 * x1 + g.wv1 * x2 - g.wv0 * x3 + g.wv0 * x4);
    gv += gq * (-g.wu1 * x1 - g.wu0 * x2 + g.wu1 * x3 + g.wu0 * x4);
    if (o1 >= 0) atomicAdd(gxc + o1, gq * w1);
    if (o2 >= 0) atomicAdd(gxc + o2, gq * w2);
    if (o3 >= 0) atomicAdd(gxc + o3, gq * w3);
    if (o4 >= 0) atomicAdd(gxc + o4, gq * w4);
  }
  float* gn = goff + (long)n * offsn;
  gn[(long)t * plane + p] = g.mu ? gu : 0.f;
  gn[(long)(9 + t) * plane + p] = g.mv ? gv : 0.f;
}

// Same contract, without floating-point atomics on gx (ds_add_f32 retires about one lane per clock: the scatter
// version keeps the LDS 100 % busy).  The sampling pattern of a tap is shared by all channels, so per (image, tap) the
// workgroup builds the TRANSPOSED sparse sampling operator once -- a CSR list, per input pixel q, of the output
// positions p and bilinear weights that touch q (counting sort with integer LDS atomics) -- and then every channel
// GATHERS: gx[c][q] += sum_{(p,w) in list(q)} w * gcol[c][t][p], each q owned by one lane.
template <int CH, int NT, bool DET>
__global__ __launch_bounds__(NT) void deform_backward_csr_kernel(const float* __restrict__ x, const float* __restrict__ off,
                                                                  const float* __restrict__ gcol,
                                                                  const float* __restrict__ w1o,
                                                                  const float* __restrict__ gy, float* __restrict__ gx,
                                                                  float* goff, int N, int C, int H, int W, long offsn) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int wtot[NT / 64];
  const int plane = H * W;
  float* sx = sm;                              // CH * plane (not in the DET variant: the offset gradients, the only
  float* sg = DET ? sm : sx + CH * plane;      // CH * plane   reader of x, are computed by deform_goff_kernel)
  int* offs = (int*)(sg + CH * plane);         // plane + 1  (counts, then exclusive offsets)
  int* cur = offs + plane + 1;                 // plane      (fill cursors)
  int* ent_p = cur + plane;                    // 4 * plane
  float* ent_w = (float*)(ent_p + 4 * plane);  // 4 * plane
  const int n = blockIdx.x, c0 = blockIdx.y * CH, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* xn = x + ((long)n * C + c0) * plane;
  for (int e = tid; e < CH * plane; e += NT) {
    if constexpr (!DET) sx[e] = xn[e];
    sg[e] = 0.f;
  }
  const float* on = off + (long)n * offsn;
  float* gn = goff + (long)n * offsn;
  const int per = (plane + NT - 1) / NT;  // elements of the scan owned by one thread
  for (int t = 0; t < 9; ++t) {
    for (int e = tid; e <= plane; e += NT) offs[e] = 0;
    __syncthreads();
    // ---- pass 1: how many samples touch each input pixel ----
    for (int p = tid; p < plane; p += NT) {
      const int a = p / W, b = p - a * W;
      const DeformGeom g = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
      const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
      const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
      if (o1 >= 0) atomicAdd(offs + o1, 1);
      if (o2 >= 0) atomicAdd(offs + o2, 1);
      if (o3 >= 0) atomicAdd(offs + o3, 1);
      if (o4 >= 0) atomicAdd(offs + o4, 1);
    }
    __syncthreads();
    // ---- exclusive scan of the counts (each thread owns `per` consecutive entries) ----
    {
      const int base = tid * per;
      int loc = 0;
      for (int i = 0; i < per; ++i)
        if (base + i < plane) loc += offs[base + i];
      int inc = loc;
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      if (lane == 63) wtot[wave] = inc;
      __syncthreads();
      int pre = inc - loc;
      for (int w2 = 0; w2 < wave; ++w2) pre += wtot[w2];
      for (int i = 0; i < per; ++i)
        if (base + i < plane) {
          const int cnt = offs[base + i];
          offs[base + i] = pre;
          cur[base + i] = pre;
          pre += cnt;
        }
      if (tid == NT - 1) offs[plane] = pre;  // the last thread owns the tail (possibly empty): pre == grand total
    }
    __syncthreads();
    // ---- pass 2: fill the lists; offset gradients (a gather already: 4 corner reads per channel) ----
    for (int p = tid; p < plane; p += NT) {
      const int a = p / W, b = p - a * W;
      const DeformGeom g = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
      const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
      const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
      const float w1 = g.wu1 * g.wv1, w2 = g.wu0 * g.wv1, w3 = g.wu1 * g.wv0, w4 = g.wu0 * g.wv0;
      if (o1 >= 0) { const int sl = atomicAdd(cur + o1, 1); ent_p[sl] = p; ent_w[sl] = w1; }
      if (o2 >= 0) { const int sl = atomicAdd(cur + o2, 1); ent_p[sl] = p; ent_w[sl] = w2; }
      if (o3 >= 0) { const int sl = atomicAdd(cur + o3, 1); ent_p[sl] = p; ent_w[sl] = w3; }
      if (o4 >= 0) { const int sl = atomicAdd(cur + o4, 1); ent_p[sl] = p; ent_w[sl] = w4; }
      if constexpr (!DET) {  // DET: the offset gradients come from deform_goff_kernel (no atomics across channel groups)
        const float gyv = gy ? gy[(long)n * plane + p] : 0.f;
        float gu = 0.f, gv = 0.f;
        float gqs[CH];
  #pragma unroll
        for (int c = 0; c < CH; ++c)
          gqs[c] = gcol ? gcol[((long)n * C * 9 + (long)(c0 + c) * 9 + t) * plane + p] : w1o[(c0 + c) * 9 + t] * gyv;
  #pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float gq = gqs[c];
          const float* xc = sx + c * plane;
          const float x1 = o1 >= 0 ? xc[o1] : 0.f, x2 = o2 >= 0 ? xc[o2] : 0.f;
          const float x3 = o3 >= 0 ? xc[o3] : 0.f, x4 = o4 >= 0 ? xc[o4] : 0.f;
          gu += gq * (-g.wv1 * x1 + g.wv1 * x2 - g.wv0 * x3 + g.wv0 * x4);
          gv += gq * (-g.wu1 * x1 - g.wu0 * x2 + g.wu1 * x3 + g.wu0 * x4);
        }
        if (g.mu) atomicAdd(gn + (long)t * plane + p, gu);
        if (g.mv) atomicAdd(gn + (long)(9 + t) * plane + p, gv);
      }
    }
    __syncthreads();
    // ---- gather: every input pixel q sums its list, channel by channel ----
    // (entries outer, channels inner: the CH gathers of one list entry are independent loads in flight together)
    for (int q = tid; q < plane; q += NT) {
      const int s0 = offs[q], s1 = offs[q + 1];
      if constexpr (DET) {  // the fill order (LDS cursor atomics) varies from run to run: sort the few entries by position
        for (int i = s0 + 1; i < s1; ++i) {
          const int kp = ent_p[i];
          const float kw = ent_w[i];
          int jj = i - 1;
          while (jj >= s0 && ent_p[jj] > kp) {
            ent_p[jj + 1] = ent_p[jj];
            ent_w[jj + 1] = ent_w[jj];
            --jj;
          }
          ent_p[jj + 1] = kp;
          ent_w[jj + 1] = kw;
        }
      }
      float acc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = 0.f;
      const float* gc0 = gcol ? gcol + ((long)n * C * 9 + (long)c0 * 9 + t) * plane : nullptr;
      for (int sl = s0; sl < s1; ++sl) {
        const int p = ent_p[sl];
        const float w = ent_w[sl];
        if (gc0) {
          float gq[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) gq[c] = gc0[(long)c * 9 * plane + p];
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[c] += w * gq[c];
        } else {
          const float gyv = w * gy[(long)n * plane + p];
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[c] += gyv * w1o[(c0 + c) * 9 + t];
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) sg[c * plane + q] += acc[c];
    }
    __syncthreads();
  }
  float* gxn = gx + ((long)n * C + c0) * plane;
  for (int e = tid; e < CH * plane; e += NT) gxn[e] = sg[e];
}

// Offset gradients without atomics (deterministic mode): one thread per (image, tap, position) walks all channels.
__global__ __launch_bounds__(256) void deform_goff_kernel(const float* __restrict__ x, const float* __restrict__ off,
                                                          const float* __restrict__ gcol, const float* __restrict__ w1o,
                                                          const float* __restrict__ gy, float* __restrict__ goff, int N, int C,
                                                          int H, int W, long offsn) {
  const int plane = H * W;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * 9 * plane) return;
  const int p = (int)(e % plane);
  const int t = (int)((e / plane) % 9);
  const int n = (int)(e / (9L * plane));
  const int a = p / W, b = p - a * W;
  const float* on = off + (long)n * offsn;
  const DeformGeom g = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
  const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
  const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
  const float* xn = x + (long)n * C * plane;
  const float gyv = gy ? gy[(long)n * plane + p] : 0.f;
  float gu = 0.f, gv = 0.f;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float* xc = xn + (long)c * plane;
    const float gq = gcol ? gcol[((long)n * C * 9 + (long)c * 9 + t) * plane + p] : w1o[c * 9 + t] * gyv;
    const float x1 = o1 >= 0 ? xc[o1] : 0.f, x2 = o2 >= 0 ? xc[o2] : 0.f;
    const float x3 = o3 >= 0 ? xc[o3] : 0.f, x4 = o4 >= 0 ? xc[o4] : 0.f;
    gu += gq * (-g.wv1 * x1 + g.wv1 * x2 - g.wv0 * x3 + g.wv0 * x4);
    gv += gq * (-g.wu1 * x1 - g.wu0 * x2 + g.wu1 * x3 + g.wu0 * x4);
  }
  float* gn = goff + (long)n * offsn;
  gn[(long)t * plane + p] = g.mu ? gu : 0.f;
  gn[(long)(9 + t) * plane + p] = g.mv ? gv : 0.f;
}

// gx is fully overwritten; goff[n][0:18] is overwritten (channels 18.. of a padded offset tensor are left alone).
void launch_deform_backward(const float* x, const float* off, const float* gcol, const float* w1o, const float* gy,
                            float* gx, float* goff, int N, int C, int H, int W, long offsn, hipStream_t s, hipStream_t aux, hipEvent_t* ev) {
  const long plane = (long)H * W;
  constexpr int CH = 8;
  const size_t lds = sizeof(float) * ((size_t)2 * CH * plane + 10 * plane + 1);
  if (lds <= 150 * 1024 && C % CH == 0) {
    // deterministic variant: x is not staged, which leaves room for 16 channels per workgroup -- one round of N * C / 16
    // workgroups, the sampling lists built half as often, sixteen gathers in flight per list entry
    constexpr int CHD = 16;
    const bool wide = C % CHD == 0 && sizeof(float) * ((size_t)CHD * plane + 10 * plane + 1) <= 150 * 1024;
    const size_t lds_det = sizeof(float) * ((size_t)(wide ? CHD : CH) * plane + 10 * plane + 1);
    static bool attr_set = false;
    if (!attr_set) {
      DBM_HIP(hipFuncSetAttribute((const void*)deform_backward_csr_kernel<CH, 1024, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
      DBM_HIP(hipFuncSetAttribute((const void*)deform_backward_csr_kernel<CH, 1024, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
      DBM_HIP(hipFuncSetAttribute((const void*)deform_backward_csr_kernel<CHD, 1024, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
      attr_set = true;
    }
    if (g_wgrad_deterministic) {
      // the offset gradients read the same inputs and write a different output: on `aux` (when the caller has a free
      // stream and two events) they run next to the gather kernel instead of behind it
      const long total = (long)N * 9 * plane;
      hipStream_t sg = s;
      if (aux && ev) {
        DBM_HIP(hipEventRecord(ev[0], s));
        DBM_HIP(hipStreamWaitEvent(aux, ev[0], 0));
        sg = aux;
      }
      hipLaunchKernelGGL(deform_goff_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sg, x, off, gcol, w1o, gy, goff, N, C,
                         H, W, offsn);
      if (sg != s) DBM_HIP(hipEventRecord(ev[1], aux));
      if (wide)
        hipLaunchKernelGGL((deform_backward_csr_kernel<CHD, 1024, true>), dim3(N, C / CHD), dim3(1024), lds_det, s, x, off, gcol, w1o,
                           gy, gx, goff, N, C, H, W, offsn);
      else
        hipLaunchKernelGGL((deform_backward_csr_kernel<CH, 1024, true>), dim3(N, C / CH), dim3(1024), lds_det, s, x, off, gcol, w1o,
                           gy, gx, goff, N, C, H, W, offsn);
      if (sg != s) DBM_HIP(hipStreamWaitEvent(s, ev[1], 0));
    } else {
      DBM_HIP(hipMemset2DAsync(goff, sizeof(float) * offsn, 0, sizeof(float) * 18 * plane, N, s));
      hipLaunchKernelGGL((deform_backward_csr_kernel<CH, 1024, false>), dim3(N, C / CH), dim3(1024), lds, s, x, off, gcol, w1o, gy, gx,
                         goff, N, C, H, W, offsn);
    }
  } else {
    DBM_HIP(hipMemsetAsync(gx, 0, sizeof(float) * N * C * plane, s));
    const long total = (long)N * 9 * plane;
    hipLaunchKernelGGL(deform_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, off, gcol, w1o,
                       gy, gx, goff, N, C, H, W, offsn);
  }
  DBM_HIP(hipGetLastError());
}

// ---- the same CSR gather in two kernels: the transposed sampling operator of an (image, tap) is built ONCE (the fused
// form above rebuilds it in each of the C / 16 workgroups of an image), stored, and a register-only kernel gathers ----
// lists of (image n, tap t): offs[(n * 9 + t) * (plane + 1) + q] .. [q + 1] delimit the entries of input pixel q in
// ent[(n * 9 + t) * 4 * plane + ...] = {output position p, bilinear weight}, sorted by p (fixed summation order).
template <int NT>
__global__ __launch_bounds__(NT) void deform_csr_build_kernel(const float* __restrict__ off, int* __restrict__ g_offs,
                                                              int2* __restrict__ g_ent, int H, int W, long offsn) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int wtot[NT / 64];
  const int plane = H * W;
  int* offs = (int*)sm;                        // plane + 1
  int* cur = offs + plane + 1;                 // plane
  int* ent_p = cur + plane;                    // 4 * plane
  float* ent_w = (float*)(ent_p + 4 * plane);  // 4 * plane
  const int n = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* on = off + (long)n * offsn;
  const int per = (plane + NT - 1) / NT;
  for (int e = tid; e <= plane; e += NT) offs[e] = 0;
  __syncthreads();
  for (int p = tid; p < plane; p += NT) {
    const int a = p / W, b = p - a * W;
    const DeformGeom g = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
    const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
    const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
    if (o1 >= 0) atomicAdd(offs + o1, 1);
    if (o2 >= 0) atomicAdd(offs + o2, 1);
    if (o3 >= 0) atomicAdd(offs + o3, 1);
    if (o4 >= 0) atomicAdd(offs + o4, 1);
  }
  __syncthreads();
  {
    const int base = tid * per;
    int loc = 0;
    for (int i = 0; i < per; ++i)
      if (base + i < plane) loc += offs[base + i];
    int inc = loc;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int pre = inc - loc;
    for (int w2 = 0; w2 < wave; ++w2) pre += wtot[w2];
    for (int i = 0; i < per; ++i)
      if (base + i < plane) {
        const int cnt = offs[base + i];
        offs[base + i] = pre;
        cur[base + i] = pre;
        pre += cnt;
      }
    if (tid == NT - 1) offs[plane] = pre;
  }
  __syncthreads();
  for (int p = tid; p < plane; p += NT) {
    const int a = p / W, b = p - a * W;
    const DeformGeom g = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
    const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
    const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
    const float w1 = g.wu1 * g.wv1, w2 = g.wu0 * g.wv1, w3 = g.wu1 * g.wv0, w4 = g.wu0 * g.wv0;
    if (o1 >= 0) { const int sl = atomicAdd(cur + o1, 1); ent_p[sl] = p; ent_w[sl] = w1; }
    if (o2 >= 0) { const int sl = atomicAdd(cur + o2, 1); ent_p[sl] = p; ent_w[sl] = w2; }
    if (o3 >= 0) { const int sl = atomicAdd(cur + o3, 1); ent_p[sl] = p; ent_w[sl] = w3; }
    if (o4 >= 0) { const int sl = atomicAdd(cur + o4, 1); ent_p[sl] = p; ent_w[sl] = w4; }
  }
  __syncthreads();
  int* go = g_offs + ((long)n * 9 + t) * (plane + 1);
  int2* ge = g_ent + ((long)n * 9 + t) * 4 * plane;
  for (int q = tid; q < plane; q += NT) {
    const int s0 = offs[q], s1 = offs[q + 1];
    for (int i = s0 + 1; i < s1; ++i) {  // the fill order varies from run to run: sort the few entries of a pixel by position
      const int kp = ent_p[i];
      const float kw = ent_w[i];
      int jj = i - 1;
      while (jj >= s0 && ent_p[jj] > kp) {
        ent_p[jj + 1] = ent_p[jj];
        ent_w[jj + 1] = ent_w[jj];
        --jj;
      }
      ent_p[jj + 1] = kp;
      ent_w[jj + 1] = kw;
    }
    go[q] = s0;
    for (int sl = s0; sl < s1; ++sl) ge[sl] = make_int2(ent_p[sl], __float_as_int(ent_w[sl]));
  }
  if (tid == 0) go[plane] = offs[plane];
}

// gx[n][c0 .. c0 + CH)[q] = sum over taps and list entries of w * gcol[c][t][p]  (or w * gy[p] * w1o[c*9+t]): input pixel q
// owned by one thread, its CH sums in registers over all nine taps, no LDS.
template <int CH, int NT>
__global__ __launch_bounds__(NT) void deform_csr_gather_kernel(const int* __restrict__ g_offs, const int2* __restrict__ g_ent,
                                                               const float* __restrict__ gcol, const float* __restrict__ w1o,
                                                               const float* __restrict__ gy, float* __restrict__ gx, int C, int plane) {
  const int n = blockIdx.x, c0 = blockIdx.y * CH;
  for (int q = threadIdx.x; q < plane; q += NT) {
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.f;
    for (int t = 0; t < 9; ++t) {
      const int* go = g_offs + ((long)n * 9 + t) * (plane + 1);
      const int2* ge = g_ent + ((long)n * 9 + t) * 4 * plane;
      const int s0 = go[q], s1 = go[q + 1];
      const float* gc0 = gcol ? gcol + (((long)n * C + c0) * 9 + t) * plane : nullptr;
      for (int sl = s0; sl < s1; ++sl) {
        const int2 en = ge[sl];
        const int p = en.x;
        const float w = __int_as_float(en.y);
        if (gc0) {
          float gq[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) gq[c] = gc0[(long)c * 9 * plane + p];
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[c] += w * gq[c];
        } else {
          const float gyv = w * gy[(long)n * plane + p];
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[c] += gyv * w1o[(c0 + c) * 9 + t];
        }
      }
    }
    float* gxn = gx + ((long)n * C + c0) * plane + q;
#pragma unroll
    for (int c = 0; c < CH; ++c) gxn[(long)c * plane] = acc[c];
  }
}

// G[n][t][q] = sum over the list entries of input pixel q of w * gy[n][p]: the transposed sampler applied to ONE value per position and
// tap (the 64 -> 1 layer's backward in premultiplied form, deform_fused.hip); one thread per (image, tap, input pixel).
__global__ __launch_bounds__(256) void deform_csr_gather1_kernel(const int* __restrict__ g_offs, const int2* __restrict__ g_ent,
                                                                 const float* __restrict__ gy, float* __restrict__ G, int plane) {
  const int n = blockIdx.z, t = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= plane) return;
  const int* go = g_offs + ((long)n * 9 + t) * (plane + 1);
  const int2* ge = g_ent + ((long)n * 9 + t) * 4 * plane;
  const float* gyn = gy + (long)n * plane;
  const int s0 = go[q], s1 = go[q + 1];
  float acc = 0.f;
  for (int sl = s0; sl < s1; ++sl) {   // (entries sorted by position: a fixed summation order)
    const int2 en = ge[sl];
    acc += __int_as_float(en.y) * gyn[en.x];
  }
  G[((long)n * 9 + t) * plane + q] = acc;
}

size_t deform_csr_workspace_floats(int N, int H, int W) {  // offsets, then the 8-byte entries (8-byte aligned)
  const size_t plane = (size_t)H * W;
  const size_t no = ((size_t)N * 9 * (plane + 1) + 1) & ~(size_t)1;
  return no + (size_t)N * 9 * 4 * plane * 2;
}

// Input gradient only: the atomic-free CSR gather above without the offset gradients (which the fused kernels of
// deform_fused.hip produce).  Returns false when a plane does not fit the kernel's LDS lists (the caller then takes
// launch_deform_backward).
bool deform_input_grad_ok(int C, int H, int W) {
  const long plane = (long)H * W;
  return C % 8 == 0 && sizeof(float) * ((size_t)8 * plane + 10 * plane + 1) <= 150 * 1024;
}

bool deform_csr_lists_ok(int C, int H, int W) {
  static const int split_env = DBM_TUNE_GETENV("DEFORM_CSR_SPLIT") ? atoi(DBM_TUNE_GETENV("DEFORM_CSR_SPLIT")) : 1;
  return split_env && C % 16 == 0 && sizeof(float) * (10 * (size_t)H * W + 1) <= 150 * 1024;
}

void launch_deform_csr_build(const float* off, float* ws, int N, int H, int W, long offsn, hipStream_t s) {
  const long plane = (long)H * W;
  DBM_CHECK(ws != nullptr && sizeof(float) * (10 * (size_t)plane + 1) <= 150 * 1024, "deformable CSR lists: plane too large");
  int* g_offs = (int*)ws;
  int2* g_ent = (int2*)(ws + (((size_t)N * 9 * (plane + 1) + 1) & ~(size_t)1));
  static bool attr2 = false;
  if (!attr2) {
    DBM_HIP(hipFuncSetAttribute((const void*)deform_csr_build_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    attr2 = true;
  }
  hipLaunchKernelGGL((deform_csr_build_kernel<1024>), dim3(N, 9), dim3(1024), sizeof(float) * (10 * (size_t)plane + 1), s, off, g_offs, g_ent,
                     H, W, offsn);
  DBM_HIP(hipGetLastError());
}

void launch_deform_input_grad(const float* x, const float* off, const float* gcol, const float* w1o, const float* gy, float* gx, int N,
                              int C, int H, int W, long offsn, hipStream_t s, float* ws, bool lists_built) {
  DBM_CHECK(deform_input_grad_ok(C, H, W), "deformable input gradient: plane too large for the CSR kernel");
  const long plane = (long)H * W;
  DBM_CHECK(!lists_built || (ws && deform_csr_lists_ok(C, H, W)), "deformable input gradient: no prebuilt lists for this shape");
  if (ws && deform_csr_lists_ok(C, H, W)) {
    // lists built once per (image, tap), then a register-only gather per (image, 16 channels)
    int* g_offs = (int*)ws;
    int2* g_ent = (int2*)(ws + (((size_t)N * 9 * (plane + 1) + 1) & ~(size_t)1));
    if (!lists_built) launch_deform_csr_build(off, ws, N, H, W, offsn, s);
    hipLaunchKernelGGL((deform_csr_gather_kernel<16, 1024>), dim3(N, C / 16), dim3(1024), 0, s, g_offs, g_ent, gcol, w1o, gy, gx, C,
                       (int)plane);
    DBM_HIP(hipGetLastError());
    return;
  }
  constexpr int CH = 8, CHD = 16;
  const bool wide = C % CHD == 0 && sizeof(float) * ((size_t)CHD * plane + 10 * plane + 1) <= 150 * 1024;
  const size_t lds = sizeof(float) * ((size_t)(wide ? CHD : CH) * plane + 10 * plane + 1);
  static bool attr_set = false;
  if (!attr_set) {
    DBM_HIP(hipFuncSetAttribute((const void*)deform_backward_csr_kernel<CH, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                152 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)deform_backward_csr_kernel<CHD, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                152 * 1024));
    attr_set = true;
  }
  if (wide)
    hipLaunchKernelGGL((deform_backward_csr_kernel<CHD, 1024, true>), dim3(N, C / CHD), dim3(1024), lds, s, x, off, gcol, w1o, gy, gx,
                       nullptr, N, C, H, W, offsn);
  else
    hipLaunchKernelGGL((deform_backward_csr_kernel<CH, 1024, true>), dim3(N, C / CH), dim3(1024), lds, s, x, off, gcol, w1o, gy, gx,
                       nullptr, N, C, H, W, offsn);
  DBM_HIP(hipGetLastError());
}

// The sampling lists of `off` (built into ws) applied to gy (N, 1, plane): G (N, 9, plane).
void launch_deform_csr_gather1(const float* off, const float* gy, float* G, int N, int H, int W, long offsn, hipStream_t s, float* ws,
                               bool lists_built) {
  const long plane = (long)H * W;
  DBM_CHECK(ws != nullptr && sizeof(float) * (10 * (size_t)plane + 1) <= 150 * 1024, "deformable CSR lists: plane too large");
  int* g_offs = (int*)ws;
  int2* g_ent = (int2*)(ws + (((size_t)N * 9 * (plane + 1) + 1) & ~(size_t)1));
  if (!lists_built) launch_deform_csr_build(off, ws, N, H, W, offsn, s);
  hipLaunchKernelGGL(deform_csr_gather1_kernel, dim3((unsigned)((plane + 255) / 256), 9, N), dim3(256), 0, s, g_offs, g_ent, gy, G, (int)plane);
  DBM_HIP(hipGetLastError());
}

// y[n][0][p] = b + sum_k w[k] * col[n][k][p]   (final_conv_layer2's 576 -> 1 GEMV, srgan_train.py:574)
__global__ __launch_bounds__(256) void gemv_cols_kernel(const float* __restrict__ col, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int N,
                                                        int K, int plane) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * plane) return;
  const int n = (int)(e / plane), p = (int)(e - (long)n * plane);
  const float* c = col + (long)n * K * plane + p;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int k = 0;
  for (; k + 3 < K; k += 4) {
    a0 = fmaf(w[k], c[(long)k * plane], a0);
    a1 = fmaf(w[k + 1], c[(long)(k + 1) * plane], a1);
    a2 = fmaf(w[k + 2], c[(long)(k + 2) * plane], a2);
    a3 = fmaf(w[k + 3], c[(long)(k + 3) * plane], a3);
  }
  for (; k < K; ++k) a0 = fmaf(w[k], c[(long)k * plane], a0);
  y[e] = (a0 + a1) + (a2 + a3) + (bias ? bias[0] : 0.f);
}

void launch_gemv_cols(const float* col, const float* w, const float* bias, float* y, int N, int K, int plane,
                      hipStream_t s) {
  const long total = (long)N * plane;
  hipLaunchKernelGGL(gemv_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, col, w, bias, y, N, K,
                     plane);
  DBM_HIP(hipGetLastError());
}

// gw[k] += sum_{n,p} gy[n][p] * col[n][k][p];  gb += sum gy     (backward of the GEMV above)
// One 1024-thread workgroup per k: one wavefront per image at a time, lanes along the plane; fixed-order tree, no
// fp32 atomics (reproducible).
__global__ __launch_bounds__(1024) void gemv_cols_wgrad_kernel(const float* __restrict__ col,
                                                               const float* __restrict__ gy, float* gw, float* gb,
                                                               int N, int K, int plane) {
  __shared__ float part[16];
  const int k = blockIdx.x;  // k == K computes the bias gradient
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int n = wave; n < N; n += 16) {
    const float* g = gy + (long)n * plane;
    const float* c = col + ((long)n * K + (k < K ? k : 0)) * plane;
    for (int p = lane; p < plane; p += 64) acc += g[p] * (k < K ? c[p] : 1.f);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < 16; ++w) v += part[w];
    if (k < K) gw[k] += v;
    else if (gb) gb[0] += v;
  }
}

void launch_gemv_cols_wgrad(const float* col, const float* gy, float* gw, float* gb, int N, int K, int plane,
                            hipStream_t s) {
  hipLaunchKernelGGL(gemv_cols_wgrad_kernel, dim3(K + 1), dim3(1024), 0, s, col, gy, gw, gb, N, K, plane);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// backward of F.resize_images(mode="nearest") x2 (srgan_train.py:556-566): 2x2 sum pool, with the
// LeakyReLU derivative of the layer below fused in (mask = that layer's retained output, or null).
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumpool2_kernel(const float* __restrict__ g, const float* __restrict__ mask,
                                                       float* __restrict__ out, long total, int H, int W, float slope) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W);
  const int y = (int)((e / W) % H);
  const long nc = e / ((long)W * H);
  const float* gp = g + (nc * 2 * H + 2 * y) * 2 * W + 2 * x;
  float v = (gp[0] + gp[1]) + (gp[2 * W] + gp[2 * W + 1]);
  if (mask) v = mask[e] >= 0.f ? v : slope * v;
  out[e] = v;
}

void launch_sumpool2(const float* g, const float* mask, float* out, long nc, int H, int W, float slope,
                     hipStream_t s) {
  const long total = nc * H * W;
  hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, mask, out, total, H, W,
                     slope);
  DBM_HIP(hipGetLastError());
}


// chainer.dataset.concat_examples on a device-resident dataset: dst row i = src row idx[i] (rows of `row` elements of V)
template <typename V>
__global__ void gather_rows_kernel(const V* __restrict__ src, V* __restrict__ dst, const int* __restrict__ idx, long row) {
  const long i = blockIdx.y;
  const V* s = src + (long)idx[i] * row;
  V* d = dst + i * row;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < row; k += (long)gridDim.x * blockDim.x) d[k] = s[k];
}

void launch_gather_rows(const void* src, void* dst, const int* d_idx, int n, size_t row_bytes, hipStream_t s) {
  const bool wide = row_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0;
  const long row = (long)(row_bytes / (wide ? 16 : 4));
  long gx = (row + 255) / 256;
  if (gx > 64) gx = 64;
  if (wide)
    hipLaunchKernelGGL(gather_rows_kernel<float4>, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, s, (const float4*)src, (float4*)dst,
                       d_idx, row);
  else
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, s, (const float*)src, (float*)dst,
                       d_idx, row);
  DBM_HIP(hipGetLastError());
}
