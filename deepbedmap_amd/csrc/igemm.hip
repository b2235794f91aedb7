// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces every L.Convolution2D call site with >= 32 input channels on the hot path
// (reference srgan_train.py:292-331, 467-503, 506-523 offset convs, 626-634) for forward,
// and -- with transposed/flipped packed weights -- the data-gradient of the same layers.
//
// Work decomposition (CDNA4): one 256-thread workgroup (4 wavefronts, one per SIMD) owns a
// 32(out-channel) x 32(output-position) tile.  Output positions are the flattened (n, a, b)
// index, so a tile may straddle images; each lane keeps its own position's gather offsets.
// The K dimension (taps x input channels) is split 4-ways across the wavefronts by input
// channel; the four partial 32x32 accumulators are reduced through 16 KB of LDS and the
// epilogue (bias, residual axpy's, LeakyReLU, gradient mask) is applied by all 256 threads with
// 128-byte coalesced stores along the position axis.
//
// MFMA operand mapping (cdna_hip_programming.md section 3): A[i = lane&31][k = lane>>5] is the
// packed weight wp[t][ci + (lane>>5)][cout0 + (lane&31)] -> a 2 x 128-byte coalesced load;
// B[k = lane>>5][j = lane&31] is x[n_j][ci + (lane>>5)][tap-shifted position j] -> gathered
// straight from global/L2 (the 9 taps re-read the same lines, so they hit the vector L1);
// D[i][j] has j = lane&31 (position) and i = (r&3) + 8*(r>>2) + 4*(lane>>5) (out channel).
// At the fp32 MFMA rate (64 cycles per instruction per SIMD) two dword loads per MFMA keep
// the L1 below half of its bandwidth, so no LDS staging of operands is needed.
#include "dbm_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void igemm_conv_kernel(const ConvDesc d) {
  __shared__ float red[4 * 1024];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int kh = lane >> 5;
  const int plane = d.OHl * d.OWl;
  const long P = (long)blockIdx.x * 32 + j;
  const bool pv = P < (long)d.N * plane;
  int n = 0, a = 0, b = 0;
  if (pv) {
    n = (int)(P / plane);
    const int r = (int)(P - (long)n * plane);
    a = r / d.OWl;
    b = r - a * d.OWl;
  }
  const int cout0 = blockIdx.y * 32;
  const int cpw = d.Cin >> 2;       // input channels per wavefront
  const int c0 = wave * cpw + kh;   // first input channel of this lane
  const int Hl = d.Hin << d.ups, Wl = d.Win << d.ups;
  const float* xn = d.x + (long)n * d.xsn + (long)c0 * d.xsc;
  const float* wlane = d.wp + (long)c0 * d.CoutP + cout0 + j;
  const long wtap = (long)d.Cin * d.CoutP;
  const long wstep = 2L * d.CoutP;
  const int iters = cpw >> 1;  // multiple of 4 because Cin % 32 == 0

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int t = 0; t < d.T; ++t) {
    const int iy = a * d.sin + d.dy[t];
    const int ix = b * d.sin + d.dx[t];
    const bool ok = pv && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
    // out-of-image taps read a zero word with stride 0: uniform control flow, no exec masking
    const float* xp = ok ? xn + ((iy >> d.ups) * d.Win + (ix >> d.ups)) : d.zeros;
    const long xstep = ok ? 2L * d.xsc : 0L;
    const float* wp = wlane + t * wtap;
    for (int s = 0; s < iters; s += 4) {
      const float b0 = xp[0], b1 = xp[xstep], b2 = xp[2 * xstep], b3 = xp[3 * xstep];
      const float a0 = wp[0], a1 = wp[wstep], a2 = wp[2 * wstep], a3 = wp[3 * wstep];
      xp += 4 * xstep;
      wp += 4 * wstep;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc, 0, 0, 0);
    }
  }

  // split-K reduction through LDS
  float* mine = red + wave * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * kh;
    mine[i * 32 + j] = acc[r];
  }
  __syncthreads();
  if (!pv) return;
  const long pix = (long)(a * d.so + d.oy0) * d.OWp + (b * d.so + d.ox0);
  const int irow = tid >> 5;  // 0..7
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = irow + 8 * q;
    const int c = cout0 + i;
    if (c >= d.Cout) continue;
    const int e = i * 32 + j;
    float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
    if (d.bias) v += d.bias[c];
    v *= d.s1;
    const long co = (long)c * d.ysc + pix;
    if (d.r1 && c < d.r1_nch) v += d.r1s * d.r1[(long)n * d.r1sn + co];
    if (d.r2) v = d.s2 * v + d.r2[(long)n * d.r2sn + co];
    float* yp = d.y + (long)n * d.ysn + co;
    if (d.accumulate) v += *yp;
    if (d.act) v = v >= 0.f ? v : d.slope * v;
    if (d.mask && c >= d.mask_c0) {
      const float m = d.mask[(long)n * d.masksn + co];
      v = m >= 0.f ? v : d.slope * v;
    }
    *yp = v;
  }
}

KernelProfiler g_profiler;

void KernelProfiler::begin(hipStream_t s, int family, double flops) {
  Rec r;
  DBM_HIP(hipEventCreate(&r.a));
  DBM_HIP(hipEventCreate(&r.b));
  r.flops = flops;
  r.family = family;
  DBM_HIP(hipEventRecord(r.a, s));
  recs.push_back(r);
}

void KernelProfiler::end(hipStream_t s) { DBM_HIP(hipEventRecord(recs.back().b, s)); }

void KernelProfiler::collect(double out[8]) {
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  for (auto& r : recs) {
    DBM_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    DBM_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    out[r.family * 3 + 0] += ms;
    out[r.family * 3 + 1] += r.flops;
    out[r.family * 3 + 2] += 1.0;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  recs.clear();
}

void launch_igemm_conv(const ConvDesc& d, hipStream_t s) {
  DBM_CHECK(d.Cin % 32 == 0, "igemm: Cin must be a multiple of 32");
  DBM_CHECK(d.CoutP % 32 == 0 && d.Cout <= d.CoutP, "igemm: bad CoutP");
  DBM_CHECK(d.T >= 1 && d.T <= DBM_MAX_TAPS, "igemm: bad tap count");
  const long total = (long)d.N * d.OHl * d.OWl;
  dim3 grid((unsigned)((total + 31) / 32), (unsigned)((d.Cout + 31) / 32));
  if (g_profiler.enabled) g_profiler.begin(s, 0, 2.0 * (double)total * d.Cout * d.Cin * d.T);
  hipLaunchKernelGGL(igemm_conv_kernel, grid, dim3(256), 0, s, d);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// weight packing
// ----------------------------------------------------------------------------------------------
struct PackDesc {
  const float* w;
  float* dst;
  int O, C, KH, KW, T, transpose, KP, MP;
  signed char ky[DBM_MAX_TAPS], kx[DBM_MAX_TAPS];
};

__global__ void pack_weights_kernel(const PackDesc p) {
  const long total = (long)p.T * p.KP * p.MP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int m = (int)(e % p.MP);
    const int k = (int)((e / p.MP) % p.KP);
    const int t = (int)(e / ((long)p.MP * p.KP));
    const int o = p.transpose ? k : m;
    const int c = p.transpose ? m : k;
    float v = 0.f;
    if (o < p.O && c < p.C) v = p.w[(((long)o * p.C + c) * p.KH + p.ky[t]) * p.KW + p.kx[t]];
    p.dst[e] = v;
  }
}

void launch_pack_weights(const float* w, int O, int C, int KH, int KW, int T, const signed char* ky,
                         const signed char* kx, int transpose, int KP, int MP, float* dst, hipStream_t s) {
  DBM_CHECK(T <= DBM_MAX_TAPS, "pack: too many taps");
  PackDesc p;
  p.w = w; p.dst = dst; p.O = O; p.C = C; p.KH = KH; p.KW = KW; p.T = T; p.transpose = transpose; p.KP = KP; p.MP = MP;
  for (int t = 0; t < T; ++t) { p.ky[t] = ky[t]; p.kx[t] = kx[t]; }
  const long total = (long)T * KP * MP;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, s, p);
  DBM_HIP(hipGetLastError());
}
