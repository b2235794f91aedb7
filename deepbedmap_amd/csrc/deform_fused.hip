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
#include <type_traits>

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
                                               int H, int W, int tid, int abl = 0, int t_lo = 0, int t_hi = 9) {
  for (int e = t_lo * DF_POS + tid; e < t_hi * DF_POS; e += 256) {
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
      if (abl) id = make_int4(base, base, base, base);   // (libdbm_measure.so only: every gather hits one cache-resident pixel)
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
                                                                  float slope, int abl) {
  __shared__ TileGeometry geo;
  __shared__ float col[2][64 * DF_LD];  // [buffer][channel][position]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const long total = (long)N * plane;
  const long P0 = (long)blockIdx.x * DF_POS;
  build_geometry(geo, off, offsn, P0, total, plane, H, W, tid, abl);
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
  // (round 5: every bias value is loaded BEFORE the first store -- vmcnt counts stores too, in order, so a load behind a store is
  //  awaited by draining the store: the interleaved form paid sixteen write round trips per thread)
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    v[r] = acc[r] + (bias ? bias[c] : 0.f);
    if (act) v[r] = v[r] >= 0.f ? v[r] : slope * v[r];
  }
  if (y) {
#pragma unroll
    for (int r = 0; r < 16; ++r) yn[(long)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * plane] = v[r];
  }
  if (yt) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(yt + Pm * 64 + ct * 32 + 8 * g + 4 * kh) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
  }
}

// ---- the 64 -> 64 layer's forward with the sampler on an LDS window, fp32 (round 6; the training tile's narrow planes) ----
// deform_conv64_fused_kernel above runs at 0.35 of the fp32 MFMA roof (111 us for 64 x 36 x 36 positions, 39 us of MFMAs): per tap it
// gathers 64 KB from global memory, blends them into an LDS tile behind a barrier and feeds the MFMAs one ds_read_b32 each.  Here (the
// idea of deform_conv64_x3w_kernel below, for planes narrow enough that a window of FULL rows fits) a workgroup of four wavefronts owns
// 128 consecutive positions of one image and stages every input row a sample with a vertical offset in [-2, 3) can touch -- <= 12 rows
// x (W + 7) pixels x 272 bytes (256 + padding: conflict-free corner reads) -- once; a wavefront owns 32 positions and BOTH
// output-channel tiles, lane (position j, k half kh) blends its position's channels of parity kh straight into the B operand: no sample
// tile, no barrier in the tap loop.  A sample whose rows leave the window (vertical offsets beyond the window; horizontally the window
// spans the whole padded row) reads global memory through the same generic pointer -- any offset is served.
// Blend expression, channel pairing per MFMA (2 cp, 2 cp + 1) and summation order (taps outer, channel pairs inner) are
// deform_conv64_fused_kernel's: the same bits.  No sample-matrix output (colout): the launcher keeps the older kernel for that.
constexpr int DWF_R = 2;
constexpr int DWF_POS = 128;
constexpr int DWF_MAXPIX = 528;               // window pixels the LDS layout is sized for (36-wide planes: 12 rows x 43 = 516)
constexpr int DWF_PIX = 272;                  // bytes between window pixels
constexpr int DWF_LDS = DWF_MAXPIX * DWF_PIX; // 143 616 bytes

__global__ __launch_bounds__(256) void deform_conv64_fusedw_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                   const float* __restrict__ wf, const float* __restrict__ bias,
                                                                   float* __restrict__ y, float* __restrict__ yt, int N, int H, int W,
                                                                   long offsn, int act, float slope, int tiles_per_image, int abl_arg) {
#ifdef DBM_MEASURE
  const int abl = abl_arg;   // (libdbm_measure.so, DBM_FUSEDW_ABL, results wrong: 1 no step loop, 2 no window staging, 8 no stores, 256 no MFMAs)
#else
  constexpr int abl = 0;
  (void)abl_arg;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char fwin[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const int n = blockIdx.x / tiles_per_image, tile = blockIdx.x - n * tiles_per_image;
  const int p0 = tile * DWF_POS, p1 = min(p0 + DWF_POS, plane) - 1;   // positions [p0, p1] of image n
  const int a0 = p0 / W, a1 = p1 / W;
  const int wy0 = a0 - 1 - DWF_R, NR = a1 - a0 + 1 + 2 * DWF_R + 3;   // window rows wy0 .. wy0 + NR - 1
  const int NC = W + 2 * DWF_R + 3, wx0 = -1 - DWF_R;                 // window columns: the whole padded row
  const int npix = NR * NC;
  const float* xn = xt + (long)n * plane * 64;
  // ---- the window: sixteen lanes per pixel (256 contiguous bytes), zeros outside the image.  ALL of a thread's pieces (<= 33) are
  // requested before the first is written: one workgroup per CU at one wavefront per SIMD has 512 registers per lane and nothing else to
  // hide a memory round trip behind (batches of eight: five round trips, 10 of a tile's 41 us) ----
  if (!(abl & 2)) {
    constexpr int NP = DWF_MAXPIX * 16 / 256;
    float4 st[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int u = it * 256 + tid, px = u >> 4, part = u & 15;
      const int hy = px / NC, hx = px - hy * NC;
      const int gy = wy0 + hy, gx = wx0 + hx;
      const bool inside = px < npix && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      st[it] = inside ? *reinterpret_cast<const float4*>(xn + ((long)gy * W + gx) * 64 + 4 * part) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int u = it * 256 + tid, px = u >> 4, part = u & 15;
      if (px < npix) *reinterpret_cast<float4*>(fwin + px * DWF_PIX + part * 16) = st[it];
    }
  }
  // ---- this lane's position, its eighteen offsets, the nine taps' geometry ----
  const int j = lane & 31, kh = lane >> 5;
  const int p = p0 + 32 * wave + j;
  const bool valid = p <= p1;
  const int a = (valid ? p : p0) / W, b = (valid ? p : p0) - a * W;
  int pg[9];        // the top-left corner of tap t in image coordinates, packed (y0 + 2) << 16 | (x0 + 2)
  float4 cw[9];     // bilinear weights (build_geometry's: a corner outside the image has weight 0)
  {
    const float* on = off + (long)n * offsn + (valid ? p : p0);
    float ox[9], oy[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      ox[t] = on[(long)t * plane];
      oy[t] = on[(long)(9 + t) * plane];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const DeformGeom q = deform_geom(ox[t], oy[t], a, b, t / 3, t % 3, H, W, 1);
      const int y0 = q.v0 - 2, x0 = q.u0 - 2;   // image coordinates of the top-left corner (deform_corner: - pad - 1)
      const bool iy0 = (unsigned)y0 < (unsigned)H, iy1 = (unsigned)(y0 + 1) < (unsigned)H;
      const bool ix0 = (unsigned)x0 < (unsigned)W, ix1 = (unsigned)(x0 + 1) < (unsigned)W;
      cw[t].x = (valid && iy0 && ix0) ? q.wu1 * q.wv1 : 0.f;
      cw[t].y = (valid && iy0 && ix1) ? q.wu0 * q.wv1 : 0.f;
      cw[t].z = (valid && iy1 && ix0) ? q.wu1 * q.wv0 : 0.f;
      cw[t].w = (valid && iy1 && ix1) ? q.wu0 * q.wv0 : 0.f;
      pg[t] = ((y0 + 2) << 16) | (x0 + 2);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  __syncthreads();

  if (!(abl & 1)) {
    // ---- 72 pipelined steps (tap, eight channels): deform_conv64_x3w_kernel's loop (see there) with fp32 MFMAs.  Step s: wait for the
    // corner pieces of step s and the weights of step s - 1; request the corner pieces of step s + 1 (lanes inside the window: LDS, the
    // others: global memory under the complementary EXEC mask); blend the eight channels as four pairs in lock step, the eight MFMAs of
    // step s - 1 (four channel pairs (2 cp, 2 cp + 1) x two output-channel tiles) dealt between the stages; request the sixteen weights
    // of step s + 1.  A lane multiplies the channels of its parity: element kh of each blended pair.
    typedef float f4t __attribute__((ext_vector_type(4)));
    typedef float f2t __attribute__((ext_vector_type(2)));
    f4t C[2][8];
    float A[4][8];          // weights of a step: [channel pair i][output-channel tile ct] = A[.][2 i + ct]; requested TWO steps ahead (one
                            // wavefront per SIMD: nobody else hides an L2 round trip -- with one step the loop took 830 cycles per
                            // step even without its MFMAs), right behind the corner pieces: ONE wait count serves taps with and
                            // without far lanes (a branch between a request and its wait makes hipcc copy registers whose loads are
                            // still in flight: wrong results, measured)
    float pb[4];            // B operands of the step before (this lane's parity of its four channel pairs)
    const unsigned lbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)fwin;
    const unsigned wvo = (unsigned)(kh * 9 * 64 + j) * 4u;   // this lane's byte offset into wf: + (((2 cp) * 9 + t) * 64 + ct * 32) * 4
    unsigned t_lds = 0, t_g1 = 0, t_g2 = 0, t_g3 = 0, t_g4 = 0;
    unsigned long t_near = ~0ul;
    auto open_tap = [&](int t) {
      const int y0 = (pg[t] >> 16) - 2, x0 = (pg[t] & 0xffff) - 2;
      const int ry = y0 - wy0, rx = x0 - wx0;
      const bool near = (unsigned)ry < (unsigned)(NR - 1);   // (0 <= rx, rx + 1 < NC always: x0 in [-2, W])
      t_lds = lbase + (near ? (ry * NC + rx) * DWF_PIX : 0);
      t_near = __builtin_amdgcn_ballot_w64(near);
      if (t_near != ~0ul) {   // (wave-uniform)
        const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1), xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
        t_g1 = (unsigned)(ya * W + xa) * 256u;
        t_g2 = (unsigned)(ya * W + xb) * 256u;
        t_g3 = (unsigned)(yb * W + xa) * 256u;
        t_g4 = (unsigned)(yb * W + xb) * 256u;
      }
    };
#define DWF_LDS_READS "ds_read_b128 %0, %[ad] offset:%[k0]\n\tds_read_b128 %1, %[ad] offset:%[k1]\n\t" \
                      "ds_read_b128 %2, %[ad]\n\tds_read_b128 %3, %[ad]\n\t" \
                      "ds_read_b128 %4, %[ad]\n\tds_read_b128 %5, %[ad]\n\t" \
                      "ds_read_b128 %6, %[ad]\n\tds_read_b128 %7, %[ad]"
    // (the second / third / fourth corner's pixel lies NC - dependent bytes further: run-time distances, folded into the address registers)
    auto corners_ks = [&](auto KS, f4t (&c)[8]) {   // c[2 corner + e]
      constexpr int ks = decltype(KS)::value;   // channels 8 ks .. 8 ks + 7: bytes ks * 32 (+ 16)
      const unsigned ad1 = t_lds, ad2 = t_lds + DWF_PIX, ad3 = t_lds + (unsigned)NC * DWF_PIX, ad4 = ad3 + DWF_PIX;
      const unsigned g1_ = t_g1, g2_ = t_g2, g3_ = t_g3, g4_ = t_g4;
      const unsigned long nm_ = t_near;
      const float* xb_ = xn;
      // ONE asm statement for both kinds of lanes (the far lanes' loads skipped by a branch INSIDE it when there are none): two
      // statements in the arms of an if would define the same registers twice, and hipcc may then copy them at the join -- before the
      // wait, while the loads are still in flight
      unsigned long sv;
      asm volatile("s_mov_b64 %[sv], exec\n\ts_and_b64 exec, %[sv], %[nm]\n\ts_cbranch_execz .Lfw_nolds%=\n\t"
                   "ds_read_b128 %0, %[a1] offset:%[o0]\n\tds_read_b128 %1, %[a1] offset:%[o1]\n\t"
                   "ds_read_b128 %2, %[a2] offset:%[o0]\n\tds_read_b128 %3, %[a2] offset:%[o1]\n\t"
                   "ds_read_b128 %4, %[a3] offset:%[o0]\n\tds_read_b128 %5, %[a3] offset:%[o1]\n\t"
                   "ds_read_b128 %6, %[a4] offset:%[o0]\n\tds_read_b128 %7, %[a4] offset:%[o1]\n"
                   ".Lfw_nolds%=:\n\t"
                   "s_andn2_b64 exec, %[sv], %[nm]\n\ts_cbranch_execz .Lfw_nofar%=\n\t"
                   "global_load_dwordx4 %0, %[g1], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %1, %[g1], %[xb] offset:%[o1]\n\t"
                   "global_load_dwordx4 %2, %[g2], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %3, %[g2], %[xb] offset:%[o1]\n\t"
                   "global_load_dwordx4 %4, %[g3], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %5, %[g3], %[xb] offset:%[o1]\n\t"
                   "global_load_dwordx4 %6, %[g4], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %7, %[g4], %[xb] offset:%[o1]\n"
                   ".Lfw_nofar%=:\n\t"
                   "s_mov_b64 exec, %[sv]"
                   : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7]), [sv] "=&s"(sv)
                   : [a1] "v"(ad1), [a2] "v"(ad2), [a3] "v"(ad3), [a4] "v"(ad4), [nm] "s"(nm_), [g1] "v"(g1_), [g2] "v"(g2_), [g3] "v"(g3_),
                     [g4] "v"(g4_), [xb] "s"(xb_), [o0] "n"(ks * 32), [o1] "n"(ks * 32 + 16)
                   : "scc");
    };
#undef DWF_LDS_READS
    auto request_corners = [&](int st, f4t (&c)[8]) {
      if ((st & 7) == 0) open_tap(st >> 3);
      switch (st & 7) {
        case 0: corners_ks(std::integral_constant<int, 0>{}, c); break;
        case 1: corners_ks(std::integral_constant<int, 1>{}, c); break;
        case 2: corners_ks(std::integral_constant<int, 2>{}, c); break;
        case 3: corners_ks(std::integral_constant<int, 3>{}, c); break;
        case 4: corners_ks(std::integral_constant<int, 4>{}, c); break;
        case 5: corners_ks(std::integral_constant<int, 5>{}, c); break;
        case 6: corners_ks(std::integral_constant<int, 6>{}, c); break;
        default: corners_ks(std::integral_constant<int, 7>{}, c); break;
      }
    };
    // the sixteen weights of step st = (tap t, channels 8 ks ..): channel pairs cp = 4 ks + i, rows ((2 cp) * 9 + t) of wf (+ kh * 9: wvo)
    auto request_weights = [&](int st, float (&aw)[8]) {
      const int t = st >> 3, ks = st & 7;
      const float* b0 = wf + ((long)(2 * (4 * ks + 0)) * 9 + t) * 64;
      const float* b1 = wf + ((long)(2 * (4 * ks + 1)) * 9 + t) * 64;
      const float* b2 = wf + ((long)(2 * (4 * ks + 2)) * 9 + t) * 64;
      const float* b3 = wf + ((long)(2 * (4 * ks + 3)) * 9 + t) * 64;
      const unsigned vo = wvo;
      asm volatile("global_load_dword %0, %[vo], %[b0]\n\tglobal_load_dword %1, %[vo], %[b0] offset:128\n\t"
                   "global_load_dword %2, %[vo], %[b1]\n\tglobal_load_dword %3, %[vo], %[b1] offset:128\n\t"
                   "global_load_dword %4, %[vo], %[b2]\n\tglobal_load_dword %5, %[vo], %[b2] offset:128\n\t"
                   "global_load_dword %6, %[vo], %[b3]\n\tglobal_load_dword %7, %[vo], %[b3] offset:128"
                   : "=&v"(aw[0]), "=&v"(aw[1]), "=&v"(aw[2]), "=&v"(aw[3]), "=&v"(aw[4]), "=&v"(aw[5]), "=&v"(aw[6]), "=&v"(aw[7])
                   : [vo] "v"(vo), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [b3] "s"(b3));
    };
#define DWF_THROUGH(c, aw) "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(aw[0]), "+v"(aw[1]), \
                           "+v"(aw[2]), "+v"(aw[3]), "+v"(aw[4]), "+v"(aw[5]), "+v"(aw[6]), "+v"(aw[7])
    // MFMA k (0..7) of the step whose weights are aw: channel pair i = k >> 1, output-channel tile ct = k & 1
    auto mfma_k = [&](int k, const float (&aw)[8]) {
      if (abl & 256) return;
      acc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[k], pb[k >> 1], acc[k & 1], 0, 0, 0);
    };
    constexpr int NST = 72;
    request_corners(0, C[0]);
    request_weights(0, A[0]);
    request_weights(1, A[1]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      f4t (&c)[8] = C[st & 1];
      float (&aw)[8] = A[(st + 3) & 3];    // the weights of step st - 1
      // Needed now: the corner pieces of step st (LDS: lgkmcnt(0); the global pieces of its far lanes, if the tap has any) and the weights
      // of step st - 1.  In issue order: ... corner pieces(st) [step st - 1], weights(st + 1) [step st - 1, right behind them]: the
      // youngest eight loads stay in flight (step 0: both weight requests of the prologue; the last step: nothing is younger).
      if (st == 0) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" : DWF_THROUGH(c, aw));
      else if (st + 1 < NST) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : DWF_THROUGH(c, aw));
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : DWF_THROUGH(c, aw));
      if (st + 1 < NST) request_corners(st + 1, C[(st + 1) & 1]);
      if (st + 2 < NST) request_weights(st + 2, A[(st + 2) & 3]);
      auto mfma = [&](int k) {
        if (st > 0) mfma_k(k, aw);
      };
      const float4 w = cw[st >> 3];
      f2t m[4];
#define DWF_FENCE asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(pb[0]), "+v"(pb[1]), "+v"(pb[2]), "+v"(pb[3]))
#define DWF_PAIR(q, k) ((f2t){c[q + (k >> 1)][2 * (k & 1)], c[q + (k >> 1)][2 * (k & 1) + 1]})   /* corner q / 2, channel pair k of the eight */
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = w.y * DWF_PAIR(2, k);
      DWF_FENCE;
      mfma(0); mfma(1);
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = __builtin_elementwise_fma((f2t){w.x, w.x}, DWF_PAIR(0, k), m[k]);
      DWF_FENCE;
      mfma(2); mfma(3);
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = __builtin_elementwise_fma((f2t){w.z, w.z}, DWF_PAIR(4, k), m[k]);
      DWF_FENCE;
      mfma(4); mfma(5);
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = __builtin_elementwise_fma((f2t){w.w, w.w}, DWF_PAIR(6, k), m[k]);
      DWF_FENCE;
      mfma(6); mfma(7);
#undef DWF_FENCE
#undef DWF_PAIR
#pragma unroll
      for (int k = 0; k < 4; ++k) pb[k] = kh ? m[k][1] : m[k][0];
    }
    {  // the last step's MFMAs
      float (&aw)[8] = A[(NST - 1) & 3];
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(aw[0]), "+v"(aw[1]), "+v"(aw[2]), "+v"(aw[3]), "+v"(aw[4]), "+v"(aw[5]), "+v"(aw[6]), "+v"(aw[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) mfma_k(k, aw);
    }
#undef DWF_THROUGH
  }
  if (!valid || (abl & 8)) return;
  const long Pm = (long)n * plane + p;
  float v[2][16];   // (all bias loads before the first store: see deform_conv64_fused_kernel)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      v[ct][r] = acc[ct][r] + (bias ? bias[c] : 0.f);
      if (act) v[ct][r] = v[ct][r] >= 0.f ? v[ct][r] : slope * v[ct][r];
    }
  if (y) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[(long)n * 64 * plane + p + (long)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * plane] = v[ct][r];
  }
  if (yt) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(yt + Pm * 64 + ct * 32 + 8 * gq + 4 * kh) =
            make_float4(v[ct][4 * gq], v[ct][4 * gq + 1], v[ct][4 * gq + 2], v[ct][4 * gq + 3]);
  }
}

// ---- weight gradient of the 64 -> 64 layer with the sampler fused in (round 6) ----
// gW[o][c][t] += sum_{n,p} gy[n][o][p] * sample(c, t, n, p);  gb[o] += sum gy.
// Until round 5 the retained forward wrote Chainer's sample matrix x_st (N, 576, plane): 191 MB at batch 64, written from the forward
// kernel's tap loop on the generator's critical path (deform64 "keep": 128 us against 112 without, 250 against 117 us INSIDE the
// iteration) and read back by a 1x1 weight-gradient GEMM on the side stream (102 us standalone, 212 MB).  This kernel re-samples instead:
// the forward kernel's sampler (corner table of the tile, sixteen lanes per 256-byte corner run, blend into a [channel][position] LDS
// tile, the next tap's gathers in flight under this tap's MFMAs) feeds MFMAs that contract over the tile's 64 POSITIONS: D[o][c] +=
// gy[o][p] * sample[c][p], one 32 x 32 accumulator tile per tap and wavefront (quadrant (o tile, c tile) = (wave & 1, wave >> 1)): nine
// tiles that live across the workgroup's tiles (persistent workgroups, strided tile assignment).  blockIdx.y = 0 / 1 takes taps 0..4 /
// 5..8: nine tiles (144 registers) beside the sampler's 80 spilled at two workgroups per CU; five fit.  Partial sums go to
// `partial` [workgroup][tap][o][c] (+ 64 bias sums), deform_wgrad64_fold_kernel adds them in workgroup order: deterministic.
__global__ __launch_bounds__(256, 2) void deform_wgrad64_fused_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                   const float* __restrict__ gy, float* __restrict__ partial, int N, int H,
                                                                   int W, long offsn, int ntiles) {
  __shared__ TileGeometry geo;
  __shared__ float col[2][64 * DF_LD];   // [buffer][channel][position]
  __shared__ float gys[64 * DF_LD];      // [out channel][position]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const long total = (long)N * plane;
  const int q = lane & 15, pi = lane >> 4;
  const float* xq = xt + 4 * q;
  const int ct = wave & 1, cq = wave >> 1, j = lane & 31, kh = lane >> 5;
  const int half = blockIdx.y;        // taps [5 half, 5 half + NT)
  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;   // (half 0, threads 0..63: the bias gradient of out channel tid)
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
  const float* asrc = gys + (ct * 32 + j) * DF_LD + kh;   // A[i = out channel][k = position parity]
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long P0 = (long)tile * DF_POS;
    __syncthreads();   // the previous tile's readers of geo / col / gys are done
    build_geometry(geo, off, offsn, P0, total, plane, H, W, tid, 0, 5 * half, half ? 9 : 5);   // (this half's taps only)
    {  // the tile of gy: wavefront w stages out channels 16 w .., 256-byte runs along the positions (positions past the end: zero)
      const long P = P0 + lane;
      const bool pv = P < total;
      const long n = pv ? P / plane : 0;
      const float* src = gy + n * 64 * plane + (pv ? P - n * plane : 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) gys[(16 * wave + i) * DF_LD + lane] = pv ? src[(long)(16 * wave + i) * plane] : 0.f;
    }
    __syncthreads();
    gather(5 * half);
    blend(col[0]);
    __syncthreads();
    if (half == 0 && tid < 64) {   // bias gradient: this tile's 64 positions of out channel tid
      float a = 0.f;
#pragma unroll 8
      for (int pl = 0; pl < DF_POS; ++pl) a += gys[tid * DF_LD + pl];
      bsum += a;
    }
    // local tap u = 0..4 (u == 4 only in half 0): tap 5 half + u, LDS buffer u & 1
    auto multiply = [&](auto U_, bool more) {
      constexpr int u = decltype(U_)::value;
      if (more) gather(5 * half + u + 1);
      __builtin_amdgcn_sched_barrier(0);  // the requests above stay in front of the MFMA block they overlap
      const float* bsrc = col[u & 1] + (cq * 32 + j) * DF_LD + kh;   // B[k = position parity][j = in channel]
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(asrc[2 * kk], bsrc[2 * kk], acc[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) blend(col[(u + 1) & 1]);
      __syncthreads();
    };
    multiply(std::integral_constant<int, 0>{}, true);
    multiply(std::integral_constant<int, 1>{}, true);
    multiply(std::integral_constant<int, 2>{}, true);
    if (half == 0) {   // (uniform per workgroup)
      multiply(std::integral_constant<int, 3>{}, true);
      multiply(std::integral_constant<int, 4>{}, false);
    } else {
      multiply(std::integral_constant<int, 3>{}, false);
    }
  }
  // ---- this workgroup's partial sums: [tap][o][c], 128-byte runs along c ----
  float* pw = partial + (size_t)blockIdx.x * (9 * 64 * 64 + 64);
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int t = 5 * half + u;
    if (t < 9) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pw[(t * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 64 + cq * 32 + j] = acc[u][r];
    }
  }
  if (half == 0 && tid < 64) pw[9 * 64 * 64 + tid] = bsum;
}

// gw (64, 64, 3, 3) += sum over the workgroups' partial tiles, in workgroup order; gb (64) likewise
__global__ __launch_bounds__(256) void deform_wgrad64_fold_kernel(const float* __restrict__ partial, int nwg, float* gw, float* gb) {
  const int e = blockIdx.x * 256 + threadIdx.x;   // index into a partial block: (t * 64 + o) * 64 + c, then 64 bias sums
  constexpr int PB = 9 * 64 * 64 + 64;
  if (e >= PB) return;
  float a = 0.f;
#pragma unroll 8
  for (int w = 0; w < nwg; ++w) a += partial[(size_t)w * PB + e];
  if (e < 9 * 64 * 64) {
    const int t = e / 4096, o = (e >> 6) & 63, c = e & 63;
    gw[(o * 64 + c) * 9 + t] += a;
  } else if (gb) {
    gb[e - 9 * 64 * 64] += a;
  }
}

// y (N, 1, plane) = bias + sum_{c,t} w[c*9+t] * sample: lane (position, channel quad) accumulates its four channels over the
// nine taps, the sixteen quads of a position fold with a fixed xor tree.
__global__ __launch_bounds__(256) void deform_conv1_fused_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int N, int H, int W, long offsn, int oc,
                                                                 int co) {
  // (oc > 1: GeneratorModel(out_channels=oc), forward only -- one launch per output channel co, w / bias already offset)
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
    if (q == 0 && P < total) {
      const long n = P / plane;
      y[(n * oc + co) * plane + (P - n * plane)] = v + (bias ? bias[0] : 0.f);
    }
  }
}

// ---- 64 -> 64 in split-bf16 arithmetic (the bf16 sweep, forward only) ----
// The same sampler; the blended samples are split x = hi + lo (hi = bf16(x), lo = bf16(x - hi)) on their way into LDS and
// the product is three v_mfma_f32_32x32x16_bf16 -- hi*hi + hi*lo + lo*hi, fp32 accumulation, 2^-16 operand precision (the
// arithmetic of conv_cl16x3_kernel, conv_cl16.hip) -- instead of 32 fp32 MFMAs per tap: 384 instead of 2048 matrix-pipe
// cycles per tap and wavefront.  What remains is the sampler's gathers (the 64 -> 1 layer's time before it was re-associated).
//   LDS sample tile: [position][64 channels] bf16, one 128-byte row per position for hi and one for lo, rows 144 bytes apart
//   (any sixteen consecutive rows -- and the lane groups of a ds_read_b128 -- then fall on distinct banks);
//   weights: wx [tap][k step][hi | lo][channel tile][lane][8] bf16 (launch_pack_deform_x3): lane = (out channel lane & 31,
//   k group lane >> 5), one coalesced 16-byte load per fragment.
typedef __bf16 dbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 dbf16x4 __attribute__((ext_vector_type(4)));
constexpr int DX_ROW = 144;                 // bytes between positions of the split sample tile
constexpr int DX_TILE = 64 * DX_ROW;        // one (hi or lo) tile

__global__ __launch_bounds__(256) void deform_conv64_x3_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                               const dbf16x8* __restrict__ wx, const float* __restrict__ bias,
                                                               float* __restrict__ y, float* __restrict__ yt, int N, int H, int W,
                                                               long offsn, int act, float slope) {
  __shared__ TileGeometry geo;
  __shared__ __attribute__((aligned(16))) unsigned char col[2][2 * DX_TILE];  // [buffer][hi | lo][position][64 bf16 (+ pad)]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const long total = (long)N * plane;
  const long P0 = (long)blockIdx.x * DF_POS;
  build_geometry(geo, off, offsn, P0, total, plane, H, W, tid);
  const int q = lane & 15, pi = lane >> 4;            // sampler role
  const float* xq = xt + 4 * q;
  const int ct = wave & 1, pt = wave >> 1, j = lane & 31, kg = lane >> 5;   // multiplier role
  const dbf16x8* wl = wx + ct * 64 + lane;            // + ((t * 4 + ks) * 2 + hl) * 128
  const int boff = (pt * 32 + j) * DX_ROW + kg * 16;  // + ks * 32 (+ DX_TILE for lo)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  __syncthreads();

  float4 c1[4], c2[4], c3[4], c4[4], cw[4];
  auto gather = [&](int t) {
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
  auto blend = [&](unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = blend4(cw[i], c1[i], c2[i], c3[i], c4[i]);
      const float f[4] = {v.x, v.y, v.z, v.w};
      dbf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (__bf16)f[e];
        lo[e] = (__bf16)(f[e] - (float)hi[e]);
      }
      unsigned char* d = dst + (16 * wave + 4 * i + pi) * DX_ROW + 8 * q;
      *reinterpret_cast<dbf16x4*>(d) = hi;
      *reinterpret_cast<dbf16x4*>(d + DX_TILE) = lo;
    }
  };
  gather(0);
  blend(col[0]);
  __syncthreads();
  auto multiply = [&](int t, bool more) {
    dbf16x8 ah[4], al[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ah[ks] = wl[((t * 4 + ks) * 2 + 0) * 128];
      al[ks] = wl[((t * 4 + ks) * 2 + 1) * 128];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) gather(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* cb = col[t & 1] + boff;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const dbf16x8 bh = *reinterpret_cast<const dbf16x8*>(cb + ks * 32);
      const dbf16x8 bl = *reinterpret_cast<const dbf16x8*>(cb + ks * 32 + DX_TILE);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc, 0, 0, 0);   // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) blend(col[(t + 1) & 1]);
    __syncthreads();
  };
  for (int t = 0; t < 8; ++t) multiply(t, true);
  multiply(8, false);
  const long Pm = P0 + pt * 32 + j;
  if (Pm >= total) return;
  const long nm = Pm / plane;
  float v[16];   // (all bias loads before the first store: see deform_conv64_fused_kernel)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
    v[r] = acc[r] + (bias ? bias[c] : 0.f);
    if (act) v[r] = v[r] >= 0.f ? v[r] : slope * v[r];
  }
  if (y) {
#pragma unroll
    for (int r = 0; r < 16; ++r) y[nm * 64 * plane + (Pm - nm * plane) + (long)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg) * plane] = v[r];
  }
  if (yt) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(yt + Pm * 64 + ct * 32 + 8 * g + 4 * kg) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
  }
}

// ---- the same layer with the sampler reading an LDS WINDOW of the input (round 6; the bf16 sweep's full-resolution planes) ----
// deform_conv64_x3_kernel gathers four corners x 256 bytes per position and tap from the vector L1: 12 GB per 1144 x 1144 plane, 0.8 of the
// 4.8 ms of a crop.  Neighbouring positions and taps share almost all of those corners as long as the offsets are a few pixels: here a
// workgroup owns a 16 x 16 tile of positions and stages the input pixels every sample with offsets in [-2, 3) can touch -- the tile, one
// ring for the 3 x 3 taps, DW_R rings for the offsets, one more row / column for the bilinear neighbour: 23 x 23 pixels x 256 bytes (fp32,
// all 64 channels) = 132 KB, once -- 2.07 pixels staged per position instead of 36 gathered.
//   * LDS layout: pixel p (row-major in the window) at p * 272: its 256 bytes + 16 of padding.  The lanes of a ds_read_b128 group hold
//     sixteen consecutive positions of a row, whose corners are (for smooth offsets) consecutive pixels: 272 bytes apart = four banks
//     further each, sixteen of them = all 64 banks.  Padding, not an XOR swizzle: a lane's 32 reads per tap (four corners x eight
//     16-byte parts) are then ONE base register + immediate offsets -- the first version spent two VALU instructions per read on swizzled
//     addresses (2 700 VALU instructions per lane and tile, more cycles than its MFMAs, LDS reads or weight loads);
//     staged through registers (LDS-DMA writes a wavefront's 1 KB contiguously: no padding).
//   * no sample tile, no geometry table, no barrier inside the tap loop: a wavefront owns two rows of the tile (32 positions, dealt to
//     the lanes by the ds_read_b128 groups as in conv_cl16.hip) and BOTH output-channel tiles; lane (j, kg) samples the eight channels
//     16 ks + 8 kg .. + 7 of its position straight into the B operand of the k step (blend in fp32, split hi / lo in registers).
//   * 36 steps (tap, k step), software-pipelined by hand (see the loop): corner pieces and weight fragments requested a step ahead, the
//     blend as four channel pairs in lock step, the previous step's MFMAs dealt between its stages.
//   * arithmetic and summation order are deform_conv64_x3_kernel's (same blend expression, same split, taps outer, k steps inner, the
//     three products small terms first): the results are the same bits.
//   * a sample that leaves the window (offsets beyond DW_R) is served INSIDE the same loop: per tap the lanes whose four corners lie in
//     the window read LDS, the others read the same bytes from global memory under the complementary EXEC mask, one step ahead like
//     everything else -- any offset is served, the window is an accelerator.  (A first version sent the whole wavefront to an
//     unpipelined loop as soon as one of its 288 samples left the window: with offsets of about a pixel -- tools/dem_model.py --
//     that is every wavefront, and the continent sweep was SLOWER than with the gathering kernel: 1.86 against 1.76 s; now 1.60.)
constexpr int DW_T = 16;                          // tile edge (positions)
constexpr int DW_R = 2;                           // offsets in [-DW_R, DW_R + 1) stay inside the window
constexpr int DW_WIN = DW_T + 2 * DW_R + 3;       // 23
constexpr int DW_NPIX = DW_WIN * DW_WIN;          // 529
constexpr int DW_PIX = 272;                       // bytes between pixels in LDS
constexpr int DW_LDS = DW_NPIX * DW_PIX;          // 143 888 bytes
constexpr int DW_STEPS = (DW_NPIX * 16 + 511) / 512;   // 16-byte pieces per thread

// half-lane h (0..31) -> (row 0/1, column 0..15) of the wavefront's 2 x 16 patch: each 16-lane group of a ds_read_b128
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) covers one row (conv_cl16.hip's patch_of)
__device__ __forceinline__ void dw_patch_of(int h, int& g, int& i) {
  if (h < 4) { g = 0; i = h; }
  else if (h < 12) { g = 1; i = h - 4; }
  else if (h < 16) { g = 0; i = h - 8; }
  else if (h < 20) { g = 1; i = h - 8; }
  else if (h < 28) { g = 0; i = h - 12; }
  else { g = 1; i = h - 16; }
}

__global__ __launch_bounds__(512) void deform_conv64_x3w_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                const dbf16x8* __restrict__ wx, const float* __restrict__ bias,
                                                                float* __restrict__ y, float* __restrict__ yt, int N, int H, int W,
                                                                long offsn, int act, float slope, int tilesX, int tilesY, int abl_arg) {
#ifdef DBM_MEASURE
  const int abl = abl_arg;   // (a run-time value costs a dozen branches per step: libdbm_measure.so's times are ~5 % above the product's)
#else
  constexpr int abl = 0;
  (void)abl_arg;
#endif
  // (abl: libdbm_measure.so only, DBM_X3W_ABL, results wrong: 1 no step loop, 2 no window staging, 4 no geometry, 8 no stores,
  //  16 every step's weight fragments are step 0's, 32 no weight loads, 64 no corner reads, 128 no blend arithmetic, 256 no MFMAs)
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup -> tile: every XCD (workgroups are dealt to the eight XCDs round robin) takes one contiguous run of tiles -- neighbouring
  // tiles share window rows and columns in that XCD's L2
  unsigned b = blockIdx.x;
  {
    const unsigned per = gridDim.x >> 3, rem = gridDim.x & 7, x = b & 7, k = b >> 3;
    b = x * per + (x < rem ? x : rem) + k;
  }
  const int tx = (int)(b % (unsigned)tilesX);
  b /= (unsigned)tilesX;
  const int ty = (int)(b % (unsigned)tilesY), n = (int)(b / (unsigned)tilesY);
  const int plane = H * W;
  const int wy0 = ty * DW_T - 1 - DW_R, wx0 = tx * DW_T - 1 - DW_R;   // image coordinates of window pixel (0, 0)
  const float* xn = xt + (long)n * plane * 64;
  if (!(abl & 2)) {  // ---- the window: sixteen lanes per pixel (256 contiguous bytes), zeros outside the image ----
    float4 st[DW_STEPS];
#pragma unroll
    for (int it = 0; it < DW_STEPS; ++it) {
      const int u = it * 512 + tid, px = u >> 4, part = u & 15;
      const int hy = px / DW_WIN, hx = px - hy * DW_WIN;
      const int gy = wy0 + hy, gx = wx0 + hx;
      const bool inside = px < DW_NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      st[it] = inside ? *reinterpret_cast<const float4*>(xn + ((long)gy * W + gx) * 64 + 4 * part) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < DW_STEPS; ++it) {
      const int u = it * 512 + tid, px = u >> 4, part = u & 15;
      if (px < DW_NPIX) *reinterpret_cast<float4*>(win + px * DW_PIX + part * 16) = st[it];
    }
  }
  // ---- this lane's position, its eighteen offsets, the nine taps' geometry ----
  int g, i;
  dw_patch_of(lane & 31, g, i);
  const int kg = lane >> 5;
  const int a = ty * DW_T + 2 * wave + g, bcol = tx * DW_T + i;
  const bool valid = a < H && bcol < W;
  const int p = a * W + bcol;
  int pg[9];        // the top-left corner of tap t in image coordinates, packed (y0 + 2) << 16 | (x0 + 2)  (y0, x0 >= -2)
  float4 cw[9];     // bilinear weights (build_geometry's: a corner outside the image has weight 0)
  if (abl & 4) {
#pragma unroll
    for (int t = 0; t < 9; ++t) { pg[t] = ((wy0 + 2 + 2) << 16) | (wx0 + 2 + 2); cw[t] = make_float4(0.f, 0.f, 0.f, 0.f); }
  } else {
    const float* on = off + (long)n * offsn + (valid ? p : 0);
    float ox[9], oy[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      ox[t] = on[(long)t * plane];
      oy[t] = on[(long)(9 + t) * plane];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const DeformGeom q = deform_geom(ox[t], oy[t], a, bcol, t / 3, t % 3, H, W, 1);
      const int y0 = q.v0 - 2, x0 = q.u0 - 2;   // image coordinates of the top-left corner (deform_corner: - pad - 1)
      const bool iy0 = (unsigned)y0 < (unsigned)H, iy1 = (unsigned)(y0 + 1) < (unsigned)H;
      const bool ix0 = (unsigned)x0 < (unsigned)W, ix1 = (unsigned)(x0 + 1) < (unsigned)W;
      cw[t].x = (valid && iy0 && ix0) ? q.wu1 * q.wv1 : 0.f;
      cw[t].y = (valid && iy0 && ix1) ? q.wu0 * q.wv1 : 0.f;
      cw[t].z = (valid && iy1 && ix0) ? q.wu1 * q.wv0 : 0.f;
      cw[t].w = (valid && iy1 && ix1) ? q.wu0 * q.wv0 : 0.f;
      pg[t] = ((y0 + 2) << 16) | (x0 + 2);
    }
  }
  const dbf16x8* wl = wx + lane;   // + ((t * 4 + ks) * 2 + hl) * 128 + ct * 64
  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  __syncthreads();

  if (!(abl & 1)) {
    // ---- 36 pipelined steps (tap, k step) ----
    // The reads are inline asm with hand-placed wait counts: left to hipcc every read ends up right in front of its first use (its IR
    // passes sink loads through __builtin_amdgcn_sched_barrier; volatile loads turn into flat loads with a wait each); the values pass
    // through the wait as "+v" operands, so that nothing using them can move above it.  Step s:
    //   wait: the corner pieces of step s and the weight fragments of step s - 1 (vmcnt / lgkmcnt count in order: everything but the
    //     four fragment loads of step s, the youngest requests);
    //   the corner pieces of step s + 1 requested (their own double buffer): lanes whose four corners lie inside the window read LDS,
    //     the others -- offsets beyond DW_R -- read the same bytes from global memory, in the same instruction slot under the
    //     complementary EXEC mask (wave-uniform: a tap without such lanes issues the LDS reads only).  Any offset is served, at the
    //     speed of its locality;
    //   blend + split of step s as four channel PAIRS in lock step, the six MFMAs of step s - 1 dealt between its stages -- a wavefront's
    //     matrix instructions run under its own vector arithmetic (left to hipcc: one pair after the other through the same two
    //     registers, eight dependent packed operations deep with a wait state between each, then six MFMAs back to back).  The empty
    //     asm statements are ordering fences: every pair's operation k before any pair's operation k + 1, MFMA k between stage k and
    //     stage k + 1 (its B operands pass through both fences);
    //   the weight fragments of step s + 1 requested into the registers the MFMAs just read.
    // Same expression trees as blend4 / the gathering kernel's split, same MFMA order per accumulator: the same bits.
    typedef float f4t __attribute__((ext_vector_type(4)));
    typedef float f2t __attribute__((ext_vector_type(2)));
    f4t C[2][8], A[2][4];
    dbf16x8 pbh, pbl;   // B operands of the step before
    const unsigned lbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)win;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(wl);   // + 4096 per step; hi ct 0 / hi ct 1 / lo ct 0 / lo ct 1: + 0 / 1024 / 2048 / 3072
    // the tap whose corners are being requested: LDS address (lanes inside the window), byte offsets of the four corners from the
    // image's first pixel (the others; a corner outside the image is clamped onto it: its weight is 0), the lanes inside
    unsigned t_lds = 0, t_g1 = 0, t_g2 = 0, t_g3 = 0, t_g4 = 0;
    unsigned long t_near = ~0ul;
    auto open_tap = [&](int t) {
      const int y0 = (pg[t] >> 16) - 2, x0 = (pg[t] & 0xffff) - 2;
      const int ry = y0 - wy0, rx = x0 - wx0;
      const bool near = (unsigned)ry < (unsigned)(DW_WIN - 1) && (unsigned)rx < (unsigned)(DW_WIN - 1);
      t_lds = lbase + (near ? (ry * DW_WIN + rx) * DW_PIX : 0) + kg * 32;
      t_near = __builtin_amdgcn_ballot_w64(near);
      if (t_near != ~0ul) {   // (wave-uniform)
        const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1), xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
        t_g1 = (unsigned)(ya * W + xa) * 256u + kg * 32;
        t_g2 = (unsigned)(ya * W + xb) * 256u + kg * 32;
        t_g3 = (unsigned)(yb * W + xa) * 256u + kg * 32;
        t_g4 = (unsigned)(yb * W + xb) * 256u + kg * 32;
      }
    };
#define DW_LDS_READS "ds_read_b128 %0, %[ad] offset:%[k0]\n\tds_read_b128 %1, %[ad] offset:%[k1]\n\t" \
                     "ds_read_b128 %2, %[ad] offset:%[k2]\n\tds_read_b128 %3, %[ad] offset:%[k3]\n\t" \
                     "ds_read_b128 %4, %[ad] offset:%[k4]\n\tds_read_b128 %5, %[ad] offset:%[k5]\n\t" \
                     "ds_read_b128 %6, %[ad] offset:%[k6]\n\tds_read_b128 %7, %[ad] offset:%[k7]"
#define DW_LDS_OFFS(KS) [k0] "n"(KS * 64), [k1] "n"(KS * 64 + 16), [k2] "n"(KS * 64 + DW_PIX), [k3] "n"(KS * 64 + DW_PIX + 16), \
                        [k4] "n"(KS * 64 + DW_WIN * DW_PIX), [k5] "n"(KS * 64 + DW_WIN * DW_PIX + 16), \
                        [k6] "n"(KS * 64 + (DW_WIN + 1) * DW_PIX), [k7] "n"(KS * 64 + (DW_WIN + 1) * DW_PIX + 16)
    auto corners_ks = [&](auto KS, f4t (&c)[8]) {   // c[2 corner + e]
      constexpr int ks = decltype(KS)::value;
      if (abl & 64) return;   // (64: no corner reads)
      // (copies: clang does not capture a variable a generic lambda names in asm operands only)
      const unsigned ad_ = t_lds, g1_ = t_g1, g2_ = t_g2, g3_ = t_g3, g4_ = t_g4;
      const unsigned long nm_ = t_near;
      const float* xb_ = xn;
      // ONE asm statement for both kinds of lanes (the far lanes' loads skipped by a branch INSIDE it when there are none): two
      // statements in the arms of an if would define the same registers twice, and hipcc may then copy them at the join -- before the
      // wait, while the loads are still in flight (it did, in deform_conv64_fusedw_kernel's first two-armed wait)
      unsigned long sv;
      asm volatile("s_mov_b64 %[sv], exec\n\ts_and_b64 exec, %[sv], %[nm]\n\ts_cbranch_execz .Lxw_nolds%=\n\t" DW_LDS_READS "\n"
                   ".Lxw_nolds%=:\n\t"
                   "s_andn2_b64 exec, %[sv], %[nm]\n\ts_cbranch_execz .Lxw_nofar%=\n\t"
                   "global_load_dwordx4 %0, %[g1], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %1, %[g1], %[xb] offset:%[o1]\n\t"
                   "global_load_dwordx4 %2, %[g2], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %3, %[g2], %[xb] offset:%[o1]\n\t"
                   "global_load_dwordx4 %4, %[g3], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %5, %[g3], %[xb] offset:%[o1]\n\t"
                   "global_load_dwordx4 %6, %[g4], %[xb] offset:%[o0]\n\tglobal_load_dwordx4 %7, %[g4], %[xb] offset:%[o1]\n"
                   ".Lxw_nofar%=:\n\t"
                   "s_mov_b64 exec, %[sv]"
                   : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7]), [sv] "=&s"(sv)
                   : [ad] "v"(ad_), DW_LDS_OFFS(ks), [nm] "s"(nm_), [g1] "v"(g1_), [g2] "v"(g2_), [g3] "v"(g3_), [g4] "v"(g4_),
                     [xb] "s"(xb_), [o0] "n"(ks * 64), [o1] "n"(ks * 64 + 16)
                   : "scc");
    };
    auto request_corners = [&](int st, f4t (&c)[8]) {
      if ((st & 3) == 0) open_tap(st >> 2);
      switch (st & 3) {
        case 0: corners_ks(std::integral_constant<int, 0>{}, c); break;
        case 1: corners_ks(std::integral_constant<int, 1>{}, c); break;
        case 2: corners_ks(std::integral_constant<int, 2>{}, c); break;
        default: corners_ks(std::integral_constant<int, 3>{}, c); break;
      }
    };
    auto request_weights = [&](int st, f4t (&aw)[4]) {
      const unsigned char* q = wp + (long)((abl & 16) ? 0 : st) * 4096;   // (16: every step multiplies by step 0's fragments)
      if (abl & 32) return;                                                  // (32: no weight loads at all)
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                   "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"
                   : "=&v"(aw[0]), "=&v"(aw[1]), "=&v"(aw[2]), "=&v"(aw[3])
                   : "v"(q));
    };
#define DW_THROUGH(c, aw) "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(aw[0]), "+v"(aw[1]), \
                          "+v"(aw[2]), "+v"(aw[3])
    // MFMA k of the step whose fragments are aw (per accumulator: hi * lo, lo * hi, hi * hi -- small terms first)
    auto mfma_k = [&](int k, const f4t (&aw)[4]) {
      const int ct = k & 1;
      const dbf16x8 av = __builtin_bit_cast(dbf16x8, aw[k < 2 ? ct : k < 4 ? 2 + ct : ct]);   // hi ct 0, hi ct 1, lo ct 0, lo ct 1
      if (abl & 256) return;   // (256: no MFMAs)
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, k < 2 ? pbl : pbh, acc[ct], 0, 0, 0);
    };
    request_corners(0, C[0]);
    request_weights(0, A[0]);
    request_weights(1, A[1]);
#pragma unroll
    for (int st = 0; st < 36; ++st) {
      f4t (&c)[8] = C[st & 1];
      f4t (&aw)[4] = A[(st + 1) & 1];    // the fragments of step st - 1
      // (in order: ... corners(st) [global part, if any], weights(st); both weight requests of the prologue are younger than corners(0),
      //  corners(1) is the youngest request at step 1)
      if (st == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : DW_THROUGH(c, aw));
      else if (st == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : DW_THROUGH(c, aw));
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : DW_THROUGH(c, aw));
      if (st + 1 < 36) request_corners(st + 1, C[(st + 1) & 1]);
      auto mfma = [&](int k) {
        if (st > 0) mfma_k(k, aw);
      };
      const float4 w = cw[st >> 2];
      dbf16x8 bh, bl;
      if (abl & 128) {   // (128: no blend / split arithmetic -- the MFMAs' B operands are raw corner bytes)
#pragma unroll
        for (int k = 0; k < 6; ++k) mfma(k);
        pbh = __builtin_bit_cast(dbf16x8, c[0]);
        pbl = __builtin_bit_cast(dbf16x8, c[1]);
        if (st >= 1 && st + 1 < 36) request_weights(st + 1, A[(st + 1) & 1]);
        continue;
      }
      // SCALAR fp32 instructions, written as asm: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- what hipcc makes of this arithmetic,
      // two channels per instruction -- do NOT run under a wavefront's own MFMAs: 8 MFMAs + 64 packed FMAs take 256 + 336 cycles, with
      // v_fma_f32 (or conversions, integer instructions) 256 + 64 (tools/experiments/ubench/mfma_valu_overlap.hip).  Volatile asm keeps
      // its order: every channel's operation k before any channel's operation k + 1; the fences pin MFMA k between two stages.
      float e[8];   // the eight channels of the step: channel 4 (k >> 2) .. of piece (k >> 2), element k & 3
#define DW_FENCE asm volatile("" : "+v"(pbh), "+v"(pbl))
#define DW_CH(q, k) c[q + (k >> 2)][k & 3]   /* corner q / 2, channel k of the eight */
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e[k]) : "v"(w.y), "v"(DW_CH(2, k)));
      DW_FENCE;
      mfma(0);
      DW_FENCE;
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e[k]) : "v"(w.x), "v"(DW_CH(0, k)));
      DW_FENCE;
      mfma(1);
      DW_FENCE;
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e[k]) : "v"(w.z), "v"(DW_CH(4, k)));
      DW_FENCE;
      mfma(2);
      DW_FENCE;
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e[k]) : "v"(w.w), "v"(DW_CH(6, k)));
      DW_FENCE;
      mfma(3);
      DW_FENCE;
      unsigned hp[4], lp[4];   // hi / lo halves, two channels per register
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hp[k]) : "v"(e[2 * k]), "v"(e[2 * k + 1]));
      DW_FENCE;
      mfma(4);
      DW_FENCE;
      float hf[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hf[2 * k]) : "v"(hp[k]));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hf[2 * k + 1]) : "v"(hp[k]));
      }
      DW_FENCE;
      mfma(5);
      DW_FENCE;
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(e[k]) : "v"(hf[k]));
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lp[k]) : "v"(e[2 * k]), "v"(e[2 * k + 1]));
#undef DW_FENCE
#undef DW_CH
      typedef unsigned u4t __attribute__((ext_vector_type(4)));
      bh = __builtin_bit_cast(dbf16x8, (u4t){hp[0], hp[1], hp[2], hp[3]});
      bl = __builtin_bit_cast(dbf16x8, (u4t){lp[0], lp[1], lp[2], lp[3]});
      pbh = bh;
      pbl = bl;
      if (st >= 1 && st + 1 < 36) request_weights(st + 1, A[(st + 1) & 1]);   // (into the registers step st - 1's MFMAs have just read)
    }
    {  // step 35's MFMAs
      f4t (&aw)[4] = A[1];
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(aw[0]), "+v"(aw[1]), "+v"(aw[2]), "+v"(aw[3]));
#pragma unroll
      for (int k = 0; k < 6; ++k) mfma_k(k, aw);
    }
#undef DW_THROUGH
#undef DW_LDS_READS
#undef DW_LDS_OFFS
  }
  // ---- epilogue: bias, LeakyReLU; the channels-last output leaves through LDS (round 6): from its accumulator layout a lane holds
  // 16-byte runs of a pixel's 256 bytes -- eight store instructions each touching 32 pixels with 32 bytes; a wavefront's 32 x 64 tile,
  // transposed through the (now free) window, leaves as whole 256-byte pixels, four per instruction ----
  if (abl & 8) return;
  float v[2][16];   // (all bias loads before the first store: see deform_conv64_fused_kernel)
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      v[ct][r] = acc[ct][r] + (bias ? bias[c] : 0.f);
      if (act) v[ct][r] = v[ct][r] >= 0.f ? v[ct][r] : slope * v[ct][r];
    }
  if (y && valid) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[(long)n * 64 * plane + p + (long)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg) * plane] = v[ct][r];
  }
  if (yt) {
    __syncthreads();   // every wavefront has finished reading the window
    float* T = reinterpret_cast<float*>(win) + wave * (32 * 68);   // [pixel j][64 channels (+ 4: rows 272 bytes apart)]
    const int jj = lane & 31;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(T + jj * 68 + ct * 32 + 8 * gq + 4 * kg) =
            make_float4(v[ct][4 * gq], v[ct][4 * gq + 1], v[ct][4 * gq + 2], v[ct][4 * gq + 3]);
    // (a wavefront reads back only what it wrote itself: no barrier)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int pj = 4 * it + (lane >> 4), part = lane & 15;
      int pg_, pi_;
      dw_patch_of(pj, pg_, pi_);
      const int pa = ty * DW_T + 2 * wave + pg_, pb_ = tx * DW_T + pi_;
      if (pa < H && pb_ < W)
        *reinterpret_cast<float4*>(yt + ((long)n * plane + (long)pa * W + pb_) * 64 + 4 * part) = *reinterpret_cast<const float4*>(T + pj * 68 + 4 * part);
    }
  }
}

// w: canonical OIHW (64, 64, 3, 3) -> wx [tap][k step][hi | lo][channel tile][lane][8]
__global__ __launch_bounds__(256) void pack_deform_x3_kernel(const float* __restrict__ w, __bf16* __restrict__ wx) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // (t, ks, ct, lane, e)
  if (idx >= 9 * 4 * 2 * 64 * 8) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, ct = (idx >> 9) & 1, ks = (idx >> 10) & 3, t = idx >> 12;
  const int o = ct * 32 + (lane & 31), c = 16 * ks + 8 * (lane >> 5) + e;
  const float v = w[((long)o * 64 + c) * 9 + t];
  const __bf16 hi = (__bf16)v;
  const __bf16 lo = (__bf16)(v - (float)hi);
  const long base = ((long)(t * 4 + ks) * 2) * 1024 + (ct * 64 + lane) * 8 + e;
  wx[base] = hi;
  wx[base + 1024] = lo;
}

// ---- the few-output-channel layer (64 -> 1: the DEM itself) with the multiplication BEFORE the sampler ----
// Bilinear sampling is linear in the sampled plane: sum_c w[c][t] * sample(x_c, pos) = sample(sum_c w[c][t] * x_c, pos).  So the
// layer is a 1x1 convolution 64 -> 9 (one plane z_t per tap, deform1_premul_kernel: reads the input once, coalesced) followed by
// nine four-corner gathers of single floats per position (deform1_sample_kernel) -- instead of gathering 9 x 4 x 256 bytes per
// position through the vector L1 (deform_conv1_fused_kernel: 12 GB of gathers for one 1144 x 1144 plane, 0.65 ms; now 0.34 + 0.05 GB).
// The sums are associated differently (channels first, then corners and taps), i.e. equal to the other order to fp32 rounding.
//
// z[(n * nz + k) * plane + p] = sum_c w[k * 64 + c'] ..., k = co * 9 + t, w in its canonical OIHW order (co, c, t)
__global__ __launch_bounds__(256) void deform1_premul_kernel(const float* __restrict__ xt, const float* __restrict__ w, float* __restrict__ z,
                                                             long total, int plane, int nz) {
  extern __shared__ __attribute__((aligned(16))) float pm[];  // wsh[nz][64] | zt[nz][64]
  float* wsh = pm;
  float* zt = pm + nz * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < nz * 64; e += 256) {  // e = k * 64 + c  <-  w[(co * 64 + c) * 9 + t]
    const int k = e >> 6, c = e & 63, co = k / 9, t = k - 9 * co;
    wsh[e] = w[((long)co * 64 + c) * 9 + t];
  }
  __syncthreads();
  const int q = lane & 15, pi = lane >> 4;
  const long P0 = (long)blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = 16 * wave + 4 * i + pi;
    const long P = P0 + pl;
    const float4 v = P < total ? *reinterpret_cast<const float4*>(xt + P * 64 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nz; ++k) {
      const float4 wr = *reinterpret_cast<const float4*>(wsh + k * 64 + 4 * q);
      float a = wr.x * v.x;
      a = fmaf(wr.y, v.y, a);
      a = fmaf(wr.z, v.z, a);
      a = fmaf(wr.w, v.w, a);
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 4, 64);
      a += __shfl_xor(a, 8, 64);
      if (q == 0) zt[k * 64 + pl] = a;
    }
  }
  __syncthreads();
  const long P = P0 + lane;
  if (P < total) {
    const long n = P / plane;
    float* dst = z + n * nz * plane + (P - n * plane);
    for (int k = wave; k < nz; k += 4) dst[(long)k * plane] = zt[k * 64 + lane];
  }
}

// The same premultiplication on the matrix pipes (round 6; nz <= 16, i.e. the model's own 64 -> 1 layer): the kernel above spends four
// FMAs, four cross-lane adds and an LDS write per (position, tap) on sixteen lanes -- 180 of the 220 us of the layer on a 1144 x 1144
// plane, against 42 us for reading the plane once.  Here z (16 rows, nz live) x 16 positions = W (16 x 64) . x (64 x 16 positions) is
// sixteen v_mfma_f32_16x16x4f32 per wavefront and 16 positions: lane (j, kq) holds row j of W for the channels kq * 16 .. + 15 (registers,
// loaded once) and reads those sixteen channels of position j as ONE 64-byte run; step s multiplies the channels {kq * 16 + s}.  fp32
// operands, fp32 accumulation; the channels are summed in another order than above (equal to fp32 rounding).
typedef float f32x4m __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void deform1_premul_mfma_kernel(const float* __restrict__ xt, const float* __restrict__ w,
                                                                  float* __restrict__ z, int total, int plane, int nz) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  float a[16];
  {
    const int co = j / 9, t = j - 9 * co;   // row j of z = (output channel co, tap t); w in its canonical OIHW order (co, c, t)
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = j < nz ? w[((long)co * 64 + kq * 16 + s) * 9 + t] : 0.f;
  }
  // A tile's 16 positions x 64 channels = 4 KB, contiguous in memory: four coalesced 1 KB loads per wavefront (lane -> 16 consecutive
  // bytes), transposed into the MFMA's B layout through a private LDS tile (rows of 272 bytes: conflict-free both ways).  Reading the
  // B layout straight from memory -- lane (j, kq) = 64 bytes at position j, sixteen such lanes 256 bytes apart -- touched 64 cache
  // lines per load instruction.
  __shared__ __attribute__((aligned(16))) float tl[4][16 * 68];
  float* mine = tl[wave];
  const long nfl = (long)total * 64;
  const int ntile = (total + 15) >> 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
    const int P = tile * 16 + j;
    float4 g[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long idx = (long)tile * 1024 + m * 256 + lane * 4;
      g[m] = idx < nfl ? *reinterpret_cast<const float4*>(xt + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) *reinterpret_cast<float4*>(mine + (4 * m + (lane >> 4)) * 68 + (lane & 15) * 4) = g[m];
    float4 b[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) b[m] = *reinterpret_cast<const float4*>(mine + j * 68 + kq * 16 + 4 * m);
    f32x4m acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + 0], b[m].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + 1], b[m].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + 2], b[m].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + 3], b[m].w, acc, 0, 0, 0);
    }
    if (P < total) {   // lane (j, kq) holds rows 4 kq .. 4 kq + 3 of column j
      const int n = P / plane, p = P - n * plane;
      float* dst = z + ((long)n * nz + 4 * kq) * plane + p;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * kq + r < nz) dst[(long)r * plane] = acc[r];
    }
  }
}

static void launch_premul(const float* xt, const float* w, float* z, long total, int plane, int nz, hipStream_t s) {
  static const int mfma_env = DBM_TUNE_GETENV("DEFORM1_PREMUL_MFMA") ? atoi(DBM_TUNE_GETENV("DEFORM1_PREMUL_MFMA")) : 1;   // (0: the vector-ALU kernel -- A/B)
  // (planes of the sweep only: on the training tile's 83 k positions the two kernels take the same time inside the iteration, and the
  //  discriminator's theoretically-zero linear_2/b gradient -- rounding noise of 128 cancelling terms, held to 3.2e-7 by
  //  tests/test_gpu_dem.py -- moves with the last bit of any fake: the training path keeps the summation order it was pinned with)
  if (nz <= 16 && mfma_env && total >= (1L << 18) && total < (1L << 31)) {
    const long ntile = (total + 15) / 16;
    const unsigned blocks = (unsigned)std::min<long>((ntile + 3) / 4, 4096);
    hipLaunchKernelGGL(deform1_premul_mfma_kernel, dim3(blocks), dim3(256), 0, s, xt, w, z, (int)total, plane, nz);
    return;
  }
  static bool attr = false;   // (O >= 15: 2 * 9 * O * 64 floats exceed the 64 KB default of dynamic LDS)
  if (!attr) {
    DBM_HIP(hipFuncSetAttribute((const void*)deform1_premul_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 9 * 16 * 64 * (int)sizeof(float)));
    attr = true;
  }
  const unsigned blocks = (unsigned)((total + DF_POS - 1) / DF_POS);
  hipLaunchKernelGGL(deform1_premul_kernel, dim3(blocks), dim3(256), (size_t)2 * nz * 64 * sizeof(float), s, xt, w, z, total, plane, nz);
}

// y[(n * oc + co) * plane + p] = bias[co] + sum_t bilinear(z[n][co * 9 + t], p + tap_t + offset_t(p));  blockIdx.y = co
__global__ __launch_bounds__(256) void deform1_sample_kernel(const float* __restrict__ z, const float* __restrict__ off,
                                                             const float* __restrict__ bias, float* __restrict__ y, long total, int H, int W,
                                                             long offsn, int oc) {
  const long P = (long)blockIdx.x * 256 + threadIdx.x;
  if (P >= total) return;
  const int plane = H * W, co = blockIdx.y;
  const int n = (int)(P / plane);
  const int p = (int)(P - (long)n * plane);
  const int a = p / W, b = p - a * W;
  const float* on = off + (long)n * offsn + p;
  const float* zn = z + ((long)n * oc + co) * 9 * plane;
  float ox[9], oy[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { ox[t] = on[(long)t * plane]; oy[t] = on[(long)(9 + t) * plane]; }
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const DeformGeom g = deform_geom(ox[t], oy[t], a, b, t / 3, t % 3, H, W, 1);
    const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
    const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
    const float* zt = zn + (long)t * plane;
    const float z1 = zt[o1 >= 0 ? o1 : 0], z2 = zt[o2 >= 0 ? o2 : 0], z3 = zt[o3 >= 0 ? o3 : 0], z4 = zt[o4 >= 0 ? o4 : 0];
    float v = (o1 >= 0 ? g.wu1 * g.wv1 : 0.f) * z1;
    v = fmaf(o2 >= 0 ? g.wu0 * g.wv1 : 0.f, z2, v);
    v = fmaf(o3 >= 0 ? g.wu1 * g.wv0 : 0.f, z3, v);
    v = fmaf(o4 >= 0 ? g.wu0 * g.wv0 : 0.f, z4, v);
    acc += v;
  }
  y[((long)n * oc + co) * plane + p] = acc + (bias ? bias[co] : 0.f);
}


// ---------------------------------------------------------------------------------------------------------------
// Backward: offset gradients (+ the pieces that share their gathers).  The corner table additionally keeps the four
// one-dimensional weights, the corner validity and the coordinate-gradient masks.
//   deform_bwd64_fused_kernel  64 -> 64 layer: dcol = W^T dy per tap on the MFMAs (M = 64 in channels, N = 64 positions,
//                              K = 64 out channels) into an LDS tile, from there (a) to the column-gradient matrix the
//                              CSR gather kernel reads and (b) into the offset gradients -- replaces a 1x1 implicit
//                              GEMM writing 191 MB and deform_goff_kernel re-reading them
//   deform_bwd1_fused_kernel   64 -> 1 layer: dcol = w (x) gy is rank one; offset gradients and, from the same
//                              gathers, the layer's weight / bias gradient as per-workgroup partial sums
//                              (deform_wgrad1_fold_kernel adds them in workgroup order) -- the sample matrix of this
//                              layer is not needed at all
// ---------------------------------------------------------------------------------------------------------------
struct TileGeometryBwd {
  int4 idx[9 * DF_POS];     // pixel index n * plane + offset of the four corners (0 where outside the image)
  float4 wuv[9 * DF_POS];   // wu0, wu1, wv0, wv1
  int flags[9 * DF_POS];    // bits 0..3: corner inside the image, bit 4: mu, bit 5: mv
};

__device__ __forceinline__ void build_geometry_bwd(TileGeometryBwd& g, const float* __restrict__ off, long offsn, long P0, long total,
                                                   int plane, int H, int W, int tid) {
  for (int e = tid; e < 9 * DF_POS; e += 256) {
    const int t = e >> 6, pl = e & 63;
    const long P = P0 + pl;
    int4 id = make_int4(0, 0, 0, 0);
    float4 wg = make_float4(0.f, 0.f, 0.f, 0.f);
    int fl = 0;
    if (P < total) {
      const int n = (int)(P / plane);
      const int p = (int)(P - (long)n * plane);
      const int a = p / W, b = p - a * W;
      const float* on = off + (long)n * offsn;
      const DeformGeom q = deform_geom(on[(long)t * plane + p], on[(long)(9 + t) * plane + p], a, b, t / 3, t % 3, H, W, 1);
      const int o1 = deform_corner(q.v0, q.u0, H, W, 1), o2 = deform_corner(q.v0, q.u0 + 1, H, W, 1);
      const int o3 = deform_corner(q.v0 + 1, q.u0, H, W, 1), o4 = deform_corner(q.v0 + 1, q.u0 + 1, H, W, 1);
      const int base = n * plane;
      if (o1 >= 0) { id.x = base + o1; fl |= 1; }
      if (o2 >= 0) { id.y = base + o2; fl |= 2; }
      if (o3 >= 0) { id.z = base + o3; fl |= 4; }
      if (o4 >= 0) { id.w = base + o4; fl |= 8; }
      if (q.mu) fl |= 16;
      if (q.mv) fl |= 32;
      wg = make_float4(q.wu0, q.wu1, q.wv0, q.wv1);
    }
    g.idx[e] = id;
    g.wuv[e] = wg;
    g.flags[e] = fl;
  }
}

__device__ __forceinline__ float4 mask4(const float4& v, bool ok) { return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f); }

// d sample / d u and d sample / d v of the four channels of a quad (deform_goff_kernel's expressions)
__device__ __forceinline__ void coord_grads(const float4& wuv, const float4& x1, const float4& x2, const float4& x3, const float4& x4,
                                            float4& du, float4& dv) {
  const float wu0 = wuv.x, wu1 = wuv.y, wv0 = wuv.z, wv1 = wuv.w;
  du.x = -wv1 * x1.x + wv1 * x2.x - wv0 * x3.x + wv0 * x4.x;
  du.y = -wv1 * x1.y + wv1 * x2.y - wv0 * x3.y + wv0 * x4.y;
  du.z = -wv1 * x1.z + wv1 * x2.z - wv0 * x3.z + wv0 * x4.z;
  du.w = -wv1 * x1.w + wv1 * x2.w - wv0 * x3.w + wv0 * x4.w;
  dv.x = -wu1 * x1.x - wu0 * x2.x + wu1 * x3.x + wu0 * x4.x;
  dv.y = -wu1 * x1.y - wu0 * x2.y + wu1 * x3.y + wu0 * x4.y;
  dv.z = -wu1 * x1.z - wu0 * x2.z + wu1 * x3.z + wu0 * x4.z;
  dv.w = -wu1 * x1.w - wu0 * x2.w + wu1 * x3.w + wu0 * x4.w;
}

__device__ __forceinline__ float quad_sum16(float v) {  // over the sixteen channel quads of a position (lanes q = lane & 15)
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// gcol (N, 576, plane) = W^T gy (row c * 9 + t), goff (N, 18.., plane)[0:18] = offset gradients.  wb = the layer's
// per-tap transposed image [t][o][c] (IgLayer::wb[1] of the layer viewed as a 1x1 convolution): coalesced A operands.
__global__ __launch_bounds__(256, 2) void deform_bwd64_fused_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                 const float* __restrict__ wb, const float* __restrict__ gy,
                                                                 float* __restrict__ gcol, float* __restrict__ goff, int N, int H, int W,
                                                                 long offsn) {
  __shared__ TileGeometryBwd geo;
  __shared__ float dys[64 * DF_LD];  // [out channel][position]
  __shared__ float dcl[2][64 * DF_LD];  // [in channel][position] of the current tap, and of the previous one (its stores go out late)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const long total = (long)N * plane;
  const long P0 = (long)blockIdx.x * DF_POS;
  build_geometry_bwd(geo, off, offsn, P0, total, plane, H, W, tid);
  {  // the tile of gy: wavefront w stages out channels 16 w .., 256-byte runs along the positions
    const long P = P0 + lane;
    const bool pv = P < total;
    const long n = pv ? P / plane : 0;
    const float* src = gy + n * 64 * plane + (pv ? P - n * plane : 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) dys[(16 * wave + i) * DF_LD + lane] = pv ? src[(long)(16 * wave + i) * plane] : 0.f;
  }
  const int q = lane & 15, pi = lane >> 4;
  const float* xq = xt + 4 * q;
  const int ct = wave & 1, pt = wave >> 1, j = lane & 31, kh = lane >> 5;
  const float* wl = wb + (long)kh * 64 + ct * 32 + j;  // + (t * 64 + 2 op) * 64
  __syncthreads();
  // B operands (gy) do not depend on the tap; they are re-read from LDS in every tap (one ds_read_b32 per MFMA): 32 registers the
  // two-deep A operand needs
  const float* bsrc = dys + kh * DF_LD + pt * 32 + j;
  const long Ps = P0 + lane;
  const bool pvs = Ps < total;
  const long nsp = pvs ? Ps / plane : 0;
  // vmcnt counts loads AND stores, in order: a load issued behind a store can only be awaited by draining the store (a write round
  // trip, ~2 us) -- the first version of this loop did that nine times per workgroup (A operand of tap t + 1 behind the stores of tap t).
  // Now the column-gradient tile is double buffered in LDS and a tap's stores are issued inside the NEXT tap, behind that tap's
  // requests (corner gathers, the A operand of the tap AFTER it: two register sets) and in front of its MFMAs; every store is a
  // branch-free buffer store (offset -1 = dropped) so that hipcc keeps exact counts, and one barrier per tap is left.
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rgc = __builtin_amdgcn_make_buffer_rsrc(gcol, 0, (int)(4L * N * 576 * plane), 0x00020000);
  const __amdgpu_buffer_rsrc_t rgo = __builtin_amdgcn_make_buffer_rsrc(goff, 0, (int)(4L * ((long)(N - 1) * offsn + 18L * plane)), 0x00020000);
  // (buffer loads: 32-bit offsets, the tap / k part in an SGPR -- the 64-bit addresses of the flat forms cost this loop 40 registers)
  const __amdgpu_buffer_rsrc_t rwb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wb), 0, 9 * 64 * 64 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xt), 0, (int)(256L * total), 0x00020000);
  const int wl_off = 4 * (kh * 64 + ct * 32 + j);
  const int gcl0 = pvs ? (int)(4 * (nsp * 576 * plane + (Ps - nsp * plane))) : -1;
  const int pstride = 4 * plane;
  int go0[4];   // byte offset of (image, tap 0, position) of this thread's four offset-gradient positions; -1: not this lane's / outside
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long P = P0 + 16 * wave + 4 * i + pi;
    const long n = P < total ? P / plane : 0;
    go0[i] = (q == 0 && P < total) ? (int)(4 * (n * offsn + (P - n * plane))) : -1;
  }
  auto load_x = [&](int pix) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rxt, pix * 256 + 16 * q, 0, 0));
  };
  float gus[4], gvs[4];
  auto stores = [&](int t) {   // tap t's column gradients (256-byte runs along the positions) and offset gradients
    const float* src = dcl[t & 1] + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = 16 * wave + i;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, src[c * DF_LD]), rgc, gcl0 >= 0 ? gcl0 + (c * 9 + t) * pstride : -1, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gus[i]), rgo, go0[i] >= 0 ? go0[i] + t * pstride : -1, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gvs[i]), rgo, go0[i] >= 0 ? go0[i] + (9 + t) * pstride : -1, 0, 0);
    }
  };
  // A operand of tap t, k pairs [16 h, 16 h + 16)
  auto load_a = [&](float (&a)[16], int t, int h) {
#pragma unroll
    for (int op = 0; op < 16; ++op)
      a[op] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rwb, wl_off, 4 * (t * 64 + 2 * (16 * h + op)) * 64, 0));
  };
  // one tap: av = the first half of its A operand (requested a whole tap ago), avn = the next tap's first half, requested here with this
  // tap's second half (which has the first sixteen MFMAs to arrive) -- two full register sets do not fit beside the corner gathers
  auto tap = [&](int t, const float (&av)[16], float (&avn)[16]) {
    float4 c1[4], c2[4], c3[4], c4[4], cw[4];
    int fl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // the corner gathers of this tap: in flight underneath the MFMAs
      const int e = t * DF_POS + 16 * wave + 4 * i + pi;
      const int4 id = geo.idx[e];
      cw[i] = geo.wuv[e];
      fl[i] = geo.flags[e];
      c1[i] = load_x(id.x);
      c2[i] = load_x(id.y);
      c3[i] = load_x(id.z);
      c4[i] = load_x(id.w);
    }
    float avh[16];
    load_a(avh, t, 1);
    load_a(avn, t < 8 ? t + 1 : 8, 0);   // (the last tap re-reads its own: nothing uses it)
    __builtin_amdgcn_sched_barrier(0);
    if (t > 0) stores(t - 1);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int op = 0; op < 16; ++op) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[op], bsrc[2 * op * DF_LD], acc, 0, 0, 0);
#pragma unroll
    for (int op = 0; op < 16; ++op) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(avh[op], bsrc[2 * (16 + op) * DF_LD], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    float* dc = dcl[t & 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) dc[(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * DF_LD + pt * 32 + j] = acc[r];
    __syncthreads();
    // offset gradients
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pl = 16 * wave + 4 * i + pi;
      const float* dq = dc + (4 * q) * DF_LD + pl;
      const float4 g4 = make_float4(dq[0], dq[DF_LD], dq[2 * DF_LD], dq[3 * DF_LD]);
      float4 du, dv;
      coord_grads(cw[i], mask4(c1[i], fl[i] & 1), mask4(c2[i], fl[i] & 2), mask4(c3[i], fl[i] & 4), mask4(c4[i], fl[i] & 8), du, dv);
      const float gu = (g4.x * du.x + g4.y * du.y) + (g4.z * du.z + g4.w * du.w);
      const float gv = (g4.x * dv.x + g4.y * dv.y) + (g4.z * dv.z + g4.w * dv.w);
      const float su = quad_sum16(gu), sv = quad_sum16(gv);
      gus[i] = (fl[i] & 16) ? su : 0.f;
      gvs[i] = (fl[i] & 32) ? sv : 0.f;
    }
  };
  float a0[16], a1[16];
  load_a(a0, 0, 0);
#pragma unroll 1
  for (int t = 0; t < 8; t += 2) {
    tap(t, a0, a1);
    tap(t + 1, a1, a0);
  }
  tap(8, a0, a1);
  stores(8);
}

// goff as above with gcol = w[c*9+t] * gy[n][p]; partial (gridDim.x, 580): this workgroup's sums of gy * sample per
// (c * 9 + t), then of gy (the bias gradient) at [576].
__global__ __launch_bounds__(256) void deform_bwd1_fused_kernel(const float* __restrict__ xt, const float* __restrict__ off,
                                                                const float* __restrict__ w, const float* __restrict__ gy,
                                                                float* __restrict__ goff, float* __restrict__ partial, int N, int H, int W,
                                                                long offsn) {
  __shared__ TileGeometryBwd geo;
  __shared__ __attribute__((aligned(16))) float wsh[9 * 64];  // [tap][channel]
  __shared__ __attribute__((aligned(16))) float red[4][9 * 64];  // per wavefront: [tap][channel] sums of gy * sample
  __shared__ float gys[DF_POS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = H * W;
  const long total = (long)N * plane;
  const long P0 = (long)blockIdx.x * DF_POS;
  build_geometry_bwd(geo, off, offsn, P0, total, plane, H, W, tid);
  for (int e = tid; e < 576; e += 256) wsh[(e % 9) * 64 + e / 9] = w[e];
  if (tid < DF_POS) gys[tid] = (P0 + tid < total) ? gy[P0 + tid] : 0.f;  // (N, 1, plane) is flat in the position index
  const int q = lane & 15, pi = lane >> 4;
  const float* xq = xt + 4 * q;
  __syncthreads();
  float gyv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gyv[i] = gys[16 * wave + 4 * i + pi];
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    float4 c1[4], c2[4], c3[4], c4[4], cw[4];
    int fl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t * DF_POS + 16 * wave + 4 * i + pi;
      const int4 id = geo.idx[e];
      cw[i] = geo.wuv[e];
      fl[i] = geo.flags[e];
      c1[i] = *reinterpret_cast<const float4*>(xq + (long)id.x * 64);
      c2[i] = *reinterpret_cast<const float4*>(xq + (long)id.y * 64);
      c3[i] = *reinterpret_cast<const float4*>(xq + (long)id.z * 64);
      c4[i] = *reinterpret_cast<const float4*>(xq + (long)id.w * 64);
    }
    const float4 wr = *reinterpret_cast<const float4*>(wsh + t * 64 + 4 * q);
    float4 ws = make_float4(0.f, 0.f, 0.f, 0.f);  // this lane's gy * sample of its four channels, over its four positions
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 x1 = mask4(c1[i], fl[i] & 1), x2 = mask4(c2[i], fl[i] & 2), x3 = mask4(c3[i], fl[i] & 4), x4 = mask4(c4[i], fl[i] & 8);
      float4 du, dv;
      coord_grads(cw[i], x1, x2, x3, x4, du, dv);
      const float g = gyv[i];
      float gu = g * ((wr.x * du.x + wr.y * du.y) + (wr.z * du.z + wr.w * du.w));
      float gv = g * ((wr.x * dv.x + wr.y * dv.y) + (wr.z * dv.z + wr.w * dv.w));
      gu = quad_sum16(gu);
      gv = quad_sum16(gv);
      const long P = P0 + 16 * wave + 4 * i + pi;
      if (q == 0 && P < total) {
        const long n = P / plane;
        float* gn = goff + n * offsn + (P - n * plane);
        gn[(long)t * plane] = (fl[i] & 16) ? gu : 0.f;
        gn[(long)(9 + t) * plane] = (fl[i] & 32) ? gv : 0.f;
      }
      const float4 sw = make_float4(cw[i].y * cw[i].w, cw[i].x * cw[i].w, cw[i].y * cw[i].z, cw[i].x * cw[i].z);  // w1..w4
      const float4 sm = blend4(sw, x1, x2, x3, x4);
      ws.x = fmaf(g, sm.x, ws.x); ws.y = fmaf(g, sm.y, ws.y); ws.z = fmaf(g, sm.z, ws.z); ws.w = fmaf(g, sm.w, ws.w);
    }
    // the four position groups of the wavefront (lanes q, q + 16, q + 32, q + 48)
    ws.x += __shfl_xor(ws.x, 16, 64); ws.y += __shfl_xor(ws.y, 16, 64); ws.z += __shfl_xor(ws.z, 16, 64); ws.w += __shfl_xor(ws.w, 16, 64);
    ws.x += __shfl_xor(ws.x, 32, 64); ws.y += __shfl_xor(ws.y, 32, 64); ws.z += __shfl_xor(ws.z, 32, 64); ws.w += __shfl_xor(ws.w, 32, 64);
    if (pi == 0) *reinterpret_cast<float4*>(&red[wave][t * 64 + 4 * q]) = ws;
  }
  __syncthreads();
  float* po = partial + (long)blockIdx.x * 580;
  for (int e = tid; e < 576; e += 256) {
    const int t = e >> 6, c = e & 63;
    po[c * 9 + t] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
  if (tid < 64) {
    float v = gys[tid];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (tid == 0) po[576] = v;
  }
}

// gw[k] += sum over workgroups of partial[wg][k], k < 576; gb[0] += ... [576].  One wavefront per output: lane l adds the
// workgroups l, l + 64, ... in order, the lanes fold with a fixed xor tree (reproducible).
__global__ __launch_bounds__(256) void deform_wgrad1_fold_kernel(const float* __restrict__ partial, int nwg, float* gw, float* gb) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k > 576) return;
  float a = 0.f;
  for (int g = lane; g < nwg; g += 64) a += partial[(long)g * 580 + k];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) {
    if (k < 576) gw[k] += a;
    else if (gb) gb[0] += a;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Round 5: the backward pass of the 64 -> 1 layer in the PREMULTIPLIED form the forward already uses (deform1_premul_kernel):
// with z_t = sum_c w[c][t] x_c (nine planes per image, kept from the forward pass) and G_t[q] = sum over the sampling
// entries (p, corner) that land on input pixel q of (bilinear weight x gy[p]) (nine planes, a CSR gather of ONE value),
//   d loss / d offset_t(p)  = gy[p] x (d bilinear z_t / d u, d v)          -- four single-float gathers per (p, t)
//   d loss / d x_c(q)       = sum_t w[c][t] G_t[q]
//   d loss / d w[c][t]      = sum_{n, q} x_c(q) G_t[q],   d loss / d b = sum gy
// -- no gather of 9 x 4 x 256 bytes per position any more (deform_bwd1_fused_kernel: 150 us at batch 64, the iteration's critical
// path) and the input-gradient gather moves 9 instead of 64 values per list entry.
// ---------------------------------------------------------------------------------------------------------------
// goff[n][t | 9 + t][p] from the forward's z planes: one thread per (position, tap)
__global__ __launch_bounds__(256) void deform1_goff_kernel(const float* __restrict__ z, const float* __restrict__ off,
                                                           const float* __restrict__ gy, float* __restrict__ goff, long total, int H, int W,
                                                           long offsn) {
  const long P = (long)blockIdx.x * 256 + threadIdx.x;
  if (P >= total) return;
  const int plane = H * W, t = blockIdx.y;
  const int n = (int)(P / plane);
  const int p = (int)(P - (long)n * plane);
  const int a = p / W, b = p - a * W;
  const float* on = off + (long)n * offsn + p;
  const DeformGeom g = deform_geom(on[(long)t * plane], on[(long)(9 + t) * plane], a, b, t / 3, t % 3, H, W, 1);
  const int o1 = deform_corner(g.v0, g.u0, H, W, 1), o2 = deform_corner(g.v0, g.u0 + 1, H, W, 1);
  const int o3 = deform_corner(g.v0 + 1, g.u0, H, W, 1), o4 = deform_corner(g.v0 + 1, g.u0 + 1, H, W, 1);
  const float* zt = z + ((long)n * 9 + t) * plane;
  const float z1 = o1 >= 0 ? zt[o1] : 0.f, z2 = o2 >= 0 ? zt[o2] : 0.f, z3 = o3 >= 0 ? zt[o3] : 0.f, z4 = o4 >= 0 ? zt[o4] : 0.f;
  // (coord_grads' expressions on the premultiplied corners)
  const float du = -g.wv1 * z1 + g.wv1 * z2 - g.wv0 * z3 + g.wv0 * z4;
  const float dv = -g.wu1 * z1 - g.wu0 * z2 + g.wu1 * z3 + g.wu0 * z4;
  const float gv = gy[P];
  float* gn = goff + (long)n * offsn + p;
  gn[(long)t * plane] = g.mu ? gv * du : 0.f;
  gn[(long)(9 + t) * plane] = g.mv ? gv * dv : 0.f;
}

// gx[n][c][q] = sum_t w[c][t] G[n][t][q]; partial (gridDim.x, 580): this workgroup's sums of x_c(q) G_t(q) per (c * 9 + t), then of gy
// at [576] (folded by deform_wgrad1_fold_kernel in workgroup order).  Lane (q16, pi) owns four channels at four positions, as in the
// kernels above; xt is the layer input channels-last.
__global__ __launch_bounds__(256) void deform1_xw_kernel(const float* __restrict__ xt, const float* __restrict__ w, const float* __restrict__ G,
                                                         const float* __restrict__ gy, float* __restrict__ gx, float* __restrict__ partial,
                                                         long total, int plane) {
  __shared__ __attribute__((aligned(16))) float wsh[9 * 64];     // [tap][channel]
  __shared__ __attribute__((aligned(16))) float red[4][9 * 64];  // per wavefront: [tap][channel] sums of x * G
  __shared__ float gts[9][DF_POS];
  __shared__ float gys[DF_POS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long P0 = (long)blockIdx.x * DF_POS;
  for (int e = tid; e < 576; e += 256) wsh[(e % 9) * 64 + e / 9] = w[e];
  for (int e = tid; e < 9 * DF_POS; e += 256) {
    const int t = e / DF_POS, pl = e - t * DF_POS;
    const long P = P0 + pl;
    float v = 0.f;
    if (P < total) {
      const long n = P / plane;
      v = G[(n * 9 + t) * plane + (P - n * plane)];
    }
    gts[t][pl] = v;
  }
  if (tid < DF_POS) gys[tid] = (P0 + tid < total) ? gy[P0 + tid] : 0.f;
  __syncthreads();
  const int q = lane & 15, pi = lane >> 4;
  float4 xv[4];
  float4 gxv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long P = P0 + 16 * wave + 4 * i + pi;
    xv[i] = P < total ? *reinterpret_cast<const float4*>(xt + P * 64 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    gxv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    const float4 wr = *reinterpret_cast<const float4*>(wsh + t * 64 + 4 * q);
    float4 ws = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = gts[t][16 * wave + 4 * i + pi];
      gxv[i].x = fmaf(wr.x, g, gxv[i].x); gxv[i].y = fmaf(wr.y, g, gxv[i].y); gxv[i].z = fmaf(wr.z, g, gxv[i].z); gxv[i].w = fmaf(wr.w, g, gxv[i].w);
      ws.x = fmaf(g, xv[i].x, ws.x); ws.y = fmaf(g, xv[i].y, ws.y); ws.z = fmaf(g, xv[i].z, ws.z); ws.w = fmaf(g, xv[i].w, ws.w);
    }
    ws.x += __shfl_xor(ws.x, 16, 64); ws.y += __shfl_xor(ws.y, 16, 64); ws.z += __shfl_xor(ws.z, 16, 64); ws.w += __shfl_xor(ws.w, 16, 64);
    ws.x += __shfl_xor(ws.x, 32, 64); ws.y += __shfl_xor(ws.y, 32, 64); ws.z += __shfl_xor(ws.z, 32, 64); ws.w += __shfl_xor(ws.w, 32, 64);
    if (pi == 0) *reinterpret_cast<float4*>(&red[wave][t * 64 + 4 * q]) = ws;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // gx is (N, 64, plane): a lane's four channels of one position
    const long P = P0 + 16 * wave + 4 * i + pi;
    if (P < total) {
      const long n = P / plane;
      float* dst = gx + (n * 64 + 4 * q) * plane + (P - n * plane);
      dst[0] = gxv[i].x; dst[plane] = gxv[i].y; dst[2L * plane] = gxv[i].z; dst[3L * plane] = gxv[i].w;
    }
  }
  __syncthreads();
  float* po = partial + (long)blockIdx.x * 580;
  for (int e = tid; e < 576; e += 256) {
    const int t = e >> 6, c = e & 63;
    po[c * 9 + t] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
  if (tid < 64) {
    float v = gys[tid];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (tid == 0) po[576] = v;
  }
}

}  // namespace

bool deform_conv_fused_ok(int C, int O) { return C == 64 && (O == 64 || (O >= 1 && O <= 16)); }

void launch_nchw_to_nhwc64(const float* x, float* xt, int N, int plane, hipStream_t s) {
  const long total = (long)N * plane;
  hipLaunchKernelGGL(nchw_to_nhwc64_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, x, xt, total, plane);
  DBM_HIP(hipGetLastError());
}

// xt: the layer input channels-last (launch_nchw_to_nhwc64, or the previous fused layer's `yt`).
// O == 64: w = the packed forward image [576][64] of the layer viewed as a 1x1 convolution over (c, tap) columns
// (IgLayer::wf, k = c * 9 + t); O == 1: w = the canonical (1, 64, 3, 3) tensor.  y (N, O, H, W) is overwritten.
// O <= 16 with a scratch `z` of N * 9 * O * H * W floats: the premultiplied form (deform1_premul_kernel + deform1_sample_kernel);
// z == nullptr: the gather-then-multiply kernel.  O == 64: y may be null when only the channels-last output yt is wanted.
void launch_deform_conv_fused(const float* xt, const float* off, const float* w, const float* bias, float* y, float* yt, float* colout,
                              int N, int C, int H, int W, long offsn, int O, int act, float slope, hipStream_t s, float* z) {
  DBM_CHECK(deform_conv_fused_ok(C, O), "fused deformable convolution: 64 input channels, 64 or <= 16 output channels");
  const long total = (long)N * H * W;
  DBM_CHECK(total < (1L << 31), "fused deformable convolution: more than 2^31 positions");
  const unsigned blocks = (unsigned)((total + DF_POS - 1) / DF_POS);
  if (g_profiler.enabled) {
    // algorithmic bytes: the NHWC input and the 18 offset planes once, the weights once, the output (and its NHWC twin / the
    // kept sampled columns of a training forward) once
    const double bytes = 4.0 * ((double)total * (C + 18 + (y ? O : 0) + (yt ? O : 0) + (colout ? 9.0 * C : 0.0)) + 9.0 * C * O);
    char tag[40];
    snprintf(tag, sizeof(tag), "deform%d_%dx%d_n%d%s", O, H, W, N, colout ? "_keep" : "");
    g_profiler.begin(s, 0, 2.0 * (double)total * O * C * 9, bytes, tag, blocks);
  }
  // (the window form: planes narrow enough that <= 128 consecutive positions + their vertical reach fit the LDS layout, no sample matrix
  //  wanted -- the same bits as the gathering kernel.  119-125 against 111 us standalone on the training tile (one workgroup of four
  //  wavefronts per CU; 704 tiles = three rounds), but 7.56-7.58 against 7.59-7.61 ms per step in three A/B series: it leaves the CUs to
  //  the kernels of the other streams.  DBM_DEFORM_FWD_WINDOW=0: the gathering kernel.  profiles/r6/ab_deform_fwd_fp32_window.txt)
  static const int fwin_env = getenv("DBM_DEFORM_FWD_WINDOW") ? atoi(getenv("DBM_DEFORM_FWD_WINDOW")) : 1;
  const int span_rows = std::min(H, (DWF_POS - 1 + W - 1) / W + 1);   // most rows 128 consecutive positions can span
  const bool fwin_ok = O == 64 && !colout && fwin_env && (span_rows + 2 * DWF_R + 3) * (W + 2 * DWF_R + 3) <= DWF_MAXPIX;
  if (fwin_ok) {
    static bool attr = false;
    if (!attr) {
      DBM_HIP(hipFuncSetAttribute((const void*)deform_conv64_fusedw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    const int tpi = (H * W + DWF_POS - 1) / DWF_POS;
    hipLaunchKernelGGL(deform_conv64_fusedw_kernel, dim3((unsigned)(N * tpi)), dim3(256), DWF_LDS, s, xt, off, w, bias, y, yt, N, H, W, offsn, act,
                       slope, tpi, DBM_MEASURE_ENV("FUSEDW_ABL"));
  } else if (O == 64)
    hipLaunchKernelGGL(deform_conv64_fused_kernel, dim3(blocks), dim3(256), 0, s, xt, off, w, bias, y, yt, colout, N, H, W, offsn, act, slope,
                       DBM_MEASURE_ENV("DEFORM_ABL"));
  else if (z) {
    const int nz = 9 * O;
    launch_premul(xt, w, z, total, H * W, nz, s);
    hipLaunchKernelGGL(deform1_sample_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)O), dim3(256), 0, s, z, off, bias, y, total, H, W,
                       offsn, O);
  } else
    for (int co = 0; co < O; ++co)  // (w: OIHW (O, 64, 3, 3))
      hipLaunchKernelGGL(deform_conv1_fused_kernel, dim3(blocks), dim3(256), 0, s, xt, off, w + (long)co * 576, bias ? bias + co : nullptr,
                         y, N, H, W, offsn, O, co);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

size_t deform_x3_packed_elems() { return (size_t)9 * 4 * 2 * 2 * 64 * 8; }
void launch_pack_deform_x3(const float* w_oihw, void* dst, hipStream_t s) {
  hipLaunchKernelGGL(pack_deform_x3_kernel, dim3((9 * 4 * 2 * 64 * 8 + 255) / 256), dim3(256), 0, s, w_oihw, (__bf16*)dst);
  DBM_HIP(hipGetLastError());
}
// the 64 -> 64 layer in split-bf16 arithmetic (forward only): wx from launch_pack_deform_x3; y and / or yt
// window: 0 the gathering kernel, 1 the LDS-window kernel, -1 the launcher's choice (DBM_DEFORM_X3_WINDOW, default: the window on planes
// with at least one workgroup per CU)
void launch_deform_conv64_x3(const float* xt, const float* off, const void* wx, const float* bias, float* y, float* yt, int N, int H, int W,
                             long offsn, int act, float slope, hipStream_t s, int window) {
  const long total = (long)N * H * W;
  DBM_CHECK(total < (1L << 31), "fused deformable convolution: more than 2^31 positions");
  const int tilesX = (W + DW_T - 1) / DW_T, tilesY = (H + DW_T - 1) / DW_T;
  const long tiles = (long)N * tilesX * tilesY;
  if (window < 0) {
    static const int env = getenv("DBM_DEFORM_X3_WINDOW") ? atoi(getenv("DBM_DEFORM_X3_WINDOW")) : 1;
    window = env && tiles >= 256 ? 1 : 0;
  }
  const unsigned blocks = window ? (unsigned)tiles : (unsigned)((total + DF_POS - 1) / DF_POS);
  if (g_profiler.enabled) {
    const double bytes = 4.0 * (double)total * (64 + 18 + (y ? 64 : 0) + (yt ? 64 : 0)) + 2.0 * 2.0 * 9 * 64 * 64;
    char tag[40];
    snprintf(tag, sizeof(tag), "deform64x3%s_%dx%d_n%d", window ? "w" : "", H, W, N);
    g_profiler.begin(s, 0, 2.0 * (double)total * 64 * 64 * 9, bytes, tag, blocks);
  }
  if (window) {
    static bool attr = false;
    if (!attr) {
      DBM_HIP(hipFuncSetAttribute((const void*)deform_conv64_x3w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(deform_conv64_x3w_kernel, dim3(blocks), dim3(512), DW_LDS, s, xt, off, (const dbf16x8*)wx, bias, y, yt, N, H, W, offsn,
                       act, slope, tilesX, tilesY, DBM_MEASURE_ENV("X3W_ABL"));
  } else
  hipLaunchKernelGGL(deform_conv64_x3_kernel, dim3(blocks), dim3(256), 0, s, xt, off, (const dbf16x8*)wx, bias, y, yt, N, H, W, offsn, act,
                     slope);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

// Backward of the 64 -> 64 layer: gcol (N, 576, H, W) and goff[:, 0:18] are overwritten (the input gradient is then
// gathered from gcol by launch_deform_input_grad).  wb = IgLayer::wb[1] of the layer's 1x1 view ([tap][o][c]).
void launch_deform_bwd64_fused(const float* xt, const float* off, const float* wb, const float* gy, float* gcol, float* goff, int N, int H,
                               int W, long offsn, hipStream_t s) {
  const long total = (long)N * H * W;
  DBM_CHECK(total < (1L << 31), "fused deformable backward: more than 2^31 positions");
  DBM_CHECK(4L * total * 576 < (1L << 31) && 4L * N * offsn < (1L << 31), "fused deformable backward: column gradients beyond 2 GB (buffer accesses)");
  const unsigned blocks = (unsigned)((total + DF_POS - 1) / DF_POS);
  if (g_profiler.enabled) {
    char tag[40];
    snprintf(tag, sizeof(tag), "deform_bwd64_%dx%d_n%d", H, W, N);  // in: x, offsets, gy, weights; out: gcol (576 planes), goff
    g_profiler.begin(s, 0, 2.0 * (double)total * 64 * 576, 4.0 * ((double)total * (64 + 18 + 64 + 576 + 18) + 576.0 * 64), tag, blocks);
  }
  hipLaunchKernelGGL(deform_bwd64_fused_kernel, dim3(blocks), dim3(256), 0, s, xt, off, wb, gy, gcol, goff, N, H, W, offsn);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

// Weight gradient of the 64 -> 64 deformable layer from the channels-last input, the offsets and gy (no sample matrix): gw (64, 64, 3, 3)
// and gb (64, may be null) are accumulated (+=) through `partial` (deform_wgrad64_partial_floats floats of scratch).
static int deform_wgrad64_wgs(long ntiles) {
  static const int n_cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
  }();
  return (int)(ntiles < n_cus ? ntiles : n_cus);   // x 2 tap halves = two resident workgroups per CU (68 KB of LDS each)
}
size_t deform_wgrad64_partial_floats(int N, int H, int W) {
  const long ntiles = ((long)N * H * W + DF_POS - 1) / DF_POS;
  return (size_t)deform_wgrad64_wgs(ntiles) * (9 * 64 * 64 + 64);
}
void launch_deform_wgrad64_fused(const float* xt, const float* off, const float* gy, float* gw, float* gb, float* partial, int N, int H, int W,
                                 long offsn, hipStream_t s) {
  const long total = (long)N * H * W;
  DBM_CHECK(total < (1L << 31), "fused deformable weight gradient: more than 2^31 positions");
  const long ntiles = (total + DF_POS - 1) / DF_POS;
  const int wgs = deform_wgrad64_wgs(ntiles);
  if (g_profiler.enabled) {
    char tag[40];
    snprintf(tag, sizeof(tag), "deform_wgrad64_%dx%d_n%d", H, W, N);  // in: x, offsets, gy; out: the partial tiles, the gradient
    g_profiler.begin(s, 1, 2.0 * (double)total * 64 * 576, 4.0 * ((double)total * (64 + 18 + 64) + 2.0 * wgs * (9 * 64 * 64 + 64) + 576.0 * 64), tag, wgs);
  }
  hipLaunchKernelGGL(deform_wgrad64_fused_kernel, dim3(wgs, 2), dim3(256), 0, s, xt, off, gy, partial, N, H, W, offsn, (int)ntiles);
  hipLaunchKernelGGL(deform_wgrad64_fold_kernel, dim3((9 * 64 * 64 + 64 + 255) / 256), dim3(256), 0, s, partial, wgs, gw, gb);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

size_t deform_bwd1_partial_floats(int N, int H, int W) { return (size_t)(((long)N * H * W + DF_POS - 1) / DF_POS) * 580; }

// Backward of the 64 -> 1 layer: goff[:, 0:18] overwritten, gw (576, OIHW of (1, 64, 3, 3)) and gb accumulated (+=) through
// `partial` (deform_bwd1_partial_floats floats of scratch).
void launch_deform_bwd1_fused(const float* xt, const float* off, const float* w, const float* gy, float* goff, float* gw, float* gb,
                              float* partial, int N, int H, int W, long offsn, hipStream_t s) {
  const long total = (long)N * H * W;
  DBM_CHECK(total < (1L << 31), "fused deformable backward: more than 2^31 positions");
  const unsigned blocks = (unsigned)((total + DF_POS - 1) / DF_POS);
  hipLaunchKernelGGL(deform_bwd1_fused_kernel, dim3(blocks), dim3(256), 0, s, xt, off, w, gy, goff, partial, N, H, W, offsn);
  hipLaunchKernelGGL(deform_wgrad1_fold_kernel, dim3(145), dim3(256), 0, s, partial, (int)blocks, gw, gb);
  DBM_HIP(hipGetLastError());
}

// Backward of the 64 -> 1 deformable layer in the premultiplied form (round 5; see deform1_goff_kernel).  z: the forward's
// premultiplied planes (N, 9, plane); Gt: scratch of N * 9 * plane floats; csr_ws: deform_csr_workspace_floats floats;
// partial: deform_bwd1_partial_floats floats.  goff (N, >= 18, plane with image stride offsn) and gx (N, 64, plane) are overwritten,
// gw (576) / gb (1) accumulated.
void launch_deform_bwd1_premul(const float* xt, const float* off, const float* w, const float* gy, const float* z, float* goff, float* gx,
                               float* gw, float* gb, float* partial, float* csr_ws, float* Gt, int N, int H, int W, long offsn,
                               hipStream_t s, bool lists_built) {
  const long plane = (long)H * W, total = (long)N * plane;
  DBM_CHECK(total < (1L << 31), "deformable backward: more than 2^31 positions");
  const unsigned blocks = (unsigned)((total + DF_POS - 1) / DF_POS);
  hipLaunchKernelGGL(deform1_goff_kernel, dim3((unsigned)((total + 255) / 256), 9), dim3(256), 0, s, z, off, gy, goff, total, H, W, offsn);
  launch_deform_csr_gather1(off, gy, Gt, N, H, W, offsn, s, csr_ws, lists_built);
  hipLaunchKernelGGL(deform1_xw_kernel, dim3(blocks), dim3(256), 0, s, xt, w, Gt, gy, gx, partial, total, (int)plane);
  hipLaunchKernelGGL(deform_wgrad1_fold_kernel, dim3(145), dim3(256), 0, s, partial, (int)blocks, gw, gb);
  DBM_HIP(hipGetLastError());
}

// z[n][t][p] = sum_c w[c][t] x_c(p) for the O == 1 layer (the forward's premultiplication alone: the op-level backward entry point)
void launch_deform1_premul(const float* xt, const float* w, float* z, int N, int H, int W, int O, hipStream_t s) {
  const long total = (long)N * H * W;
  DBM_CHECK(O >= 1 && O <= 16, "launch_deform1_premul: up to sixteen output channels");
  launch_premul(xt, w, z, total, H * W, 9 * O, s);
  DBM_HIP(hipGetLastError());
}
