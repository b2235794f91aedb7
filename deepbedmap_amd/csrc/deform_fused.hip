// Deformable convolution with the sampler fused into the GEMM (reference srgan_train.py:506-523, :572-574:
// L.DeformableConvolution2D(64 -> 64 | 1, ksize 3, pad 1) on the 4x upsampled planes).
//
// Chainer materialises x_st = the (N, 64*9, H, W) matrix of bilinear samples and multiplies it with W.  At batch 64
// that matrix is 191 MB per layer (3 GB for one 288 x 288 inference crop); deform_sample_kernel + a 1x1 implicit GEMM
// wrote and re-read it.  Here a workgroup samples ONE tap of 64 channels x 64 positions into LDS (16 KB, double
// buffered) and feeds the fp32 MFMAs from there; in a pass nobody differentiates the samples never exist in memory.
//
//   nchw_to_nhwc64_kernel        the layer input once more in channels-last order (what the sampler gathers from)
//   deform_conv64_fused_kernel   64 -> 64 (+ bias, LeakyReLU): MFMA, M = 64 out channels, N = 64 positions, K = 9 x 64
//   deform_conv1_fused_kernel    64 -> 1  (+ bias): a dot product per position, no LDS tile
//
// What bounds a sampler is the number of gather INSTRUCTIONS (the texture addresser spends 20-30 cycles on a wave-wide
// dword gather: 3 M of them per layer at batch 64 = the 83 us of deform_sample_kernel, however little they fetch).
// From a channels-last copy the 64 channels of a corner are one contiguous 256-byte run: sixteen lanes fetch it with
// one 16-byte load each, a wave instruction covers four (position, corner) runs, and a tap costs a quarter of the
// instructions, all of them fully coalesced.  The corner offsets and bilinear weights of the tile's 64 positions x 9
// taps are computed once per workgroup into an LDS table (the fp32 normalise / denormalise round trip of Chainer's
// sampler costs two divisions per coordinate).
//
// Roles inside a 256-thread workgroup: SAMPLER -- lane = (position lane >> 4, channel quad lane & 15), wavefront w
// owns positions 16 w .. 16 w + 15 in four steps; MULTIPLIER (64 -> 64) -- wavefront w owns the 32 x 32 output tile
// (channels 32 (w & 1) .., positions 32 (w >> 1) ..) over the whole K.  The gathers of tap t + 1 are issued BEFORE
// the MFMAs of tap t and blended / written to LDS after them.
#include "dbm_internal.h"
#include "kernels.h"
#include "deform_geom.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int DF_POS = 64;   // positions per workgroup
constexpr int DF_LD = 65;    // LDS row stride of the sample tile [channel][position] (odd: conflict-free column access)

// x (N, 64, plane) -> xt (N * plane, 64)
__global__ __launch_bounds__(256) void nchw_to_nhwc64_kernel(const float* __restrict__ x, float* __restrict__ xt, long total, int plane) {
  __shared__ float tile[64 * DF_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long P0 = (long)blockIdx.x * 64;
  {
    const long P = P0 + lane;
    if (P < total) {
      const long n = P / plane;
      const float* src = x + (n * 64) * plane + (P - n * plane);
#pragma unroll
      for (int i = 0; i < 16; ++i) tile[(16 * wave + i) * DF_LD + lane] = src[(long)(16 * wave + i) * plane];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long P = P0 + 16 * wave + i;
    if (P < total) xt[P * 64 + lane] = tile[lane * DF_LD + 16 * wave + i];
  }
}

// Corner table of a 64-position tile: for (tap t, position pl) the four pixel indices n * plane + offset into the
// channels-last copy and the four bilinear weights; a corner outside the image keeps index 0 with weight 0.
struct TileGeometry {
  int4 idx[9 * DF_POS];
  float4 wgt[9 * DF_POS];
};

__device__ __forceinline__ void build_geometry(TileGeometry& g, const float* __restrict__ off, long offsn, long P0, long total, int plane,
                                               int H, int W, int tid) {
  for (int e = tid; e < 9 * DF_POS; e += 256) {
    const int t = e >> 6, pl = e & 63;
    const long P = P0 + pl;
    int4 id = make_int4(0, 0, 0, 0);
    float4 wg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P < total) {
      const int n = (int)(P / plane);
      const int p = (int)(P - (long)n * plane);
      const int a = p / W, b = p - a * W;
      const float* on = off + (long)n * offsn;
      const DeformGeom q = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
      const int o1 = deform_corner(q.v0, q.u0, H, W, 1), o2 = deform_corner(q.v0, q.u0 + 1, H, W, 1);
      const int o3 = deform_corner(q.v0 + 1, q.u0, H, W, 1), o4 = deform_corner(q.v0 + 1, q.u0 + 1, H, W, 1);
      const int base = n * plane;
      if (o1 >= 0) { id.x = base + o1; wg.x = q.wu1 * q.wv1; }
      if (o2 >= 0) { id.y = base + o2; wg.y = q.wu0 * q.wv1; }
      if (o3 >= 0) { id.z = base + o3; wg.z = q.wu1 * q.wv0; }
      if (o4 >= 0) { id.w = base + o4; wg.w = q.wu0 * q.wv0; }
    }
    g.idx[e] = id;
    g.wgt[e] = wg;
  }
}

__device__ __forceinline__ float4 blend4(const float4& w, const float4& a, const float4& b, const float4& c, const float4& d) {
  float4 r;
  r.x = w.x * a.x + w.y * b.x + w.z * c.x + w.w * d.x;
  r.y = w.x * a.y + w.y * b.y + w.z * c.y + w.w * d.y;
  r.z = w.x * a.z + w.y * b.z + w.z * c.z + w.w * d.z;
  r.w = w.x * a.w + w.y * b.w + w.z * c.w + w.w * d.w;
  return r;
}

// y (N, 64, plane) = act(bias + W * samples); yt (optional): the same output channels-last (the next deformable layer's
// sampler input); colout (optional): the samples, (N, 576, plane) with row c * 9 + t (a retained pass: the weight
// gradient reads them).
__global__ __launch_bounds__(256) void deform_conv64_fused_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                  const float* __restrict__ wf, const float* __restrict__ bias,
                                                                  float* __restrict__ y, float* __restrict__ yt,
                                                                  float* __restrict__ colout, int N, int H, int W, long offsn, int act,
                                                                  float slope) {
  __shared__ TileGeometry geo;
  __shared__ float col[2][64 * DF_LD];  // [buffer][channel][position]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const long total = (long)N * plane;
  const long P0 = (long)blockIdx.x * DF_POS;
  build_geometry(geo, off, offsn, P0, total, plane, H, W, tid);
  // ---- sampler role ----
  const int q = lane & 15, pi = lane >> 4;
  const float* xq = xt + 4 * q;
  // ---- multiplier role ----
  const int ct = wave & 1, pt = wave >> 1, j = lane & 31, kh = lane >> 5;
  const float* wl = wf + (long)kh * 9 * 64 + ct * 32 + j;  // + ((2 cp) * 9 + t) * 64
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  __syncthreads();

  float4 c1[4], c2[4], c3[4], c4[4], cw[4];
  auto gather = [&](int t) {  // requests the four corners of this lane's channel quad at its four positions
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t * DF_POS + 16 * wave + 4 * i + pi;
      const int4 id = geo.idx[e];
      cw[i] = geo.wgt[e];
      c1[i] = *reinterpret_cast<const float4*>(xq + (long)id.x * 64);
      c2[i] = *reinterpret_cast<const float4*>(xq + (long)id.y * 64);
      c3[i] = *reinterpret_cast<const float4*>(xq + (long)id.z * 64);
      c4[i] = *reinterpret_cast<const float4*>(xq + (long)id.w * 64);
    }
  };
  auto blend = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = blend4(cw[i], c1[i], c2[i], c3[i], c4[i]);
      float* d = dst + (4 * q) * DF_LD + 16 * wave + 4 * i + pi;
      d[0] = v.x; d[DF_LD] = v.y; d[2 * DF_LD] = v.z; d[3 * DF_LD] = v.w;
    }
  };
  auto spill = [&](int t, const float* src) {  // the tap's samples to the column matrix: 256-byte runs along the positions
    const long P = P0 + lane;
    if (P >= total) return;
    const long n = P / plane;
    float* dst = colout + (n * 576 + t) * plane + (P - n * plane);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = 16 * wave + i;
      dst[(long)c * 9 * plane] = src[c * DF_LD + lane];
    }
  };
  gather(0);
  blend(col[0]);
  __syncthreads();
  auto multiply = [&](int t, bool more) {
    // this tap's weights first, then the next tap's gathers: loads return in order, so the MFMAs wait for the (L2-hot,
    // coalesced) weights only and the gathers stay in flight underneath them.  (`more` is a compile-time constant at
    // both call sites: a branch around the gathers would make the waitcnt bookkeeping at its join assume the worst.)
    float av[32];
#pragma unroll
    for (int cp = 0; cp < 32; ++cp) av[cp] = wl[(long)(cp * 18 + t) * 64];
    __builtin_amdgcn_sched_barrier(0);
    if (more) gather(t + 1);
    __builtin_amdgcn_sched_barrier(0);  // the requests above stay in front of the MFMA block they overlap
    const float* cb = col[t & 1] + kh * DF_LD + pt * 32 + j;
#pragma unroll
    for (int cp = 0; cp < 32; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cp], cb[cp * 2 * DF_LD], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (colout) spill(t, col[t & 1]);
    if (more) blend(col[(t + 1) & 1]);
    __syncthreads();
  };
  for (int t = 0; t < 8; ++t) multiply(t, true);
  multiply(8, false);
  // ---- epilogue: bias, LeakyReLU; 128-byte runs along the position axis (+ 16-byte channel quads, channels-last) ----
  const long Pm = P0 + pt * 32 + j;
  if (Pm >= total) return;
  const long nm = Pm / plane;
  float* yn = y + nm * 64 * plane + (Pm - nm * plane);
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    v[r] = acc[r] + (bias ? bias[c] : 0.f);
    if (act) v[r] = v[r] >= 0.f ? v[r] : slope * v[r];
    yn[(long)c * plane] = v[r];
  }
  if (yt) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(yt + Pm * 64 + ct * 32 + 8 * g + 4 * kh) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
  }
}

// y (N, 1, plane) = bias + sum_{c,t} w[c*9+t] * sample: lane (position, channel quad) accumulates its four channels over the
// nine taps, the sixteen quads of a position fold with a fixed xor tree.
__global__ __launch_bounds__(256) void deform_conv1_fused_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int N, int H, int W, long offsn) {
  __shared__ TileGeometry geo;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const long total = (long)N * plane;
  const long P0 = (long)blockIdx.x * DF_POS;
  build_geometry(geo, off, offsn, P0, total, plane, H, W, tid);
  const int q = lane & 15, pi = lane >> 4;
  const float* xq = xt + 4 * q;
  __shared__ __attribute__((aligned(16))) float wsh[9 * 64];  // [tap][channel]
  for (int e = tid; e < 576; e += 256) wsh[(e % 9) * 64 + e / 9] = w[e];
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    float4 c1[4], c2[4], c3[4], c4[4], cw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t * DF_POS + 16 * wave + 4 * i + pi;
      const int4 id = geo.idx[e];
      cw[i] = geo.wgt[e];
      c1[i] = *reinterpret_cast<const float4*>(xq + (long)id.x * 64);
      c2[i] = *reinterpret_cast<const float4*>(xq + (long)id.y * 64);
      c3[i] = *reinterpret_cast<const float4*>(xq + (long)id.z * 64);
      c4[i] = *reinterpret_cast<const float4*>(xq + (long)id.w * 64);
    }
    const float4 wr = *reinterpret_cast<const float4*>(wsh + t * 64 + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = blend4(cw[i], c1[i], c2[i], c3[i], c4[i]);
      acc[i] = fmaf(wr.x, v.x, acc[i]);
      acc[i] = fmaf(wr.y, v.y, acc[i]);
      acc[i] = fmaf(wr.z, v.z, acc[i]);
      acc[i] = fmaf(wr.w, v.w, acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v = acc[i];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    const long P = P0 + 16 * wave + 4 * i + pi;
    if (q == 0 && P < total) y[P] = v + (bias ? bias[0] : 0.f);
  }
}

}  // namespace

bool deform_conv_fused_ok(int C, int O) { return C == 64 && (O == 64 || O == 1); }

void launch_nchw_to_nhwc64(const float* x, float* xt, int N, int plane, hipStream_t s) {
  const long total = (long)N * plane;
  hipLaunchKernelGGL(nchw_to_nhwc64_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, x, xt, total, plane);
  DBM_HIP(hipGetLastError());
}

// xt: the layer input channels-last (launch_nchw_to_nhwc64, or the previous fused layer's `yt`).
// O == 64: w = the packed forward image [576][64] of the layer viewed as a 1x1 convolution over (c, tap) columns
// (IgLayer::wf, k = c * 9 + t); O == 1: w = the canonical (1, 64, 3, 3) tensor.  y (N, O, H, W) is overwritten.
void launch_deform_conv_fused(const float* xt, const float* off, const float* w, const float* bias, float* y, float* yt, float* colout,
                              int N, int C, int H, int W, long offsn, int O, int act, float slope, hipStream_t s) {
  DBM_CHECK(deform_conv_fused_ok(C, O), "fused deformable convolution: 64 input channels, 64 or 1 output channels");
  const long total = (long)N * H * W;
  DBM_CHECK(total < (1L << 31), "fused deformable convolution: more than 2^31 positions");
  const unsigned blocks = (unsigned)((total + DF_POS - 1) / DF_POS);
  if (g_profiler.enabled) g_profiler.begin(s, 0, 2.0 * (double)total * O * C * 9);
  if (O == 64)
    hipLaunchKernelGGL(deform_conv64_fused_kernel, dim3(blocks), dim3(256), 0, s, xt, off, w, bias, y, yt, colout, N, H, W, offsn, act, slope);
  else
    hipLaunchKernelGGL(deform_conv1_fused_kernel, dim3(blocks), dim3(256), 0, s, xt, off, w, bias, y, N, H, W, offsn);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}
