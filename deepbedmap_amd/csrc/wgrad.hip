// Weight-gradient of the convolutions (backward of the L.Convolution2D call sites listed in
// igemm.hip) on the fp32 matrix cores:  gW[o][c][t] += scale * sum_{n,a,b} dy[n][o][a][b] * x[n][c][tap t].
//
// GEMM view: M = out channels, N = (in channel, tap), K = output positions.  The contiguous
// memory axis (positions) is the MFMA K axis, which per-lane global loads cannot feed
// efficiently (every lane would touch its own cache line), so operands are staged through LDS:
// a workgroup stages, for one "band" (IB images x R output rows), the 32 x BP slab of dy and
// the zero-padded, tap-ready input patch of up to 128 input channels, then each wavefront owns
// one 32-channel input tile and keeps T (= taps) independent 32x32 accumulators, i.e. T
// independent MFMA chains fed by 1 + 2/T LDS dwords per MFMA.  Bands are strided over
// gridDim.z workgroups; partial sums are folded into the canonical OIHW gradient with fp32
// atomics once per workgroup (the gradient arena is zeroed by cleargrads()).
#include "dbm_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradPlan {
  WgradDesc d;
  int G;        // 32-channel input tiles per workgroup (= wavefronts per workgroup)
  int IB, R;    // band = IB images x R output rows
  int nbr;      // row-bands per image
  int BP, BPp;  // positions per band, padded to even
  int YS;       // LDS row stride of the dy slab (odd)
  int Rin, Wst; // staged logical input patch rows / cols per image
  int ImgS;     // Rin*Wst
  int XS;       // LDS channel stride of the patch (odd)
  int nbands;
};

template <int T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradPlan p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WgradDesc& d = p.d;
  float* ldsY = lds;                          // 32 * YS
  int* pixoff = (int*)(ldsY + 32 * p.YS);     // BPp
  float* ldsX = (float*)(pixoff + p.BPp);     // G*32 * XS
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int cout0 = blockIdx.y * 32;
  const int cin0 = blockIdx.x * p.G * 32;
  const int cin_w = cin0 + wave * 32;          // this wavefront's input tile
  const bool wave_active = cin_w < d.Cin;
  const int Hl = d.Hin << d.ups, Wl = d.Win << d.ups;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // position -> patch offset table (band shape is the same for every band; the last row-band of
  // an image may be short, handled by zeroing dy there)
  for (int e = tid; e < p.BPp; e += nthr) {
    int off = 0;
    if (e < p.BP) {
      const int ib = e / (p.R * d.OW);
      const int rem = e - ib * (p.R * d.OW);
      const int al = rem / d.OW, b = rem - al * d.OW;
      off = ib * p.ImgS + al * d.stride * p.Wst + b * d.stride;
    }
    pixoff[e] = off;
  }

  for (int band = blockIdx.z; band < p.nbands; band += gridDim.z) {
    const int ig = band / p.nbr;            // image group
    const int a0 = (band - ig * p.nbr) * p.R;
    const int n0 = ig * p.IB;
    __syncthreads();  // previous band fully consumed
    // ---- stage dy slab: 32 x BPp ----
    for (int e = tid; e < 32 * p.BPp; e += nthr) {
      const int i = e / p.BPp, pix = e - i * p.BPp;
      float v = 0.f;
      if (pix < p.BP && cout0 + i < d.Cout) {
        const int ib = pix / (p.R * d.OW);
        const int rem = pix - ib * (p.R * d.OW);
        const int al = rem / d.OW, b = rem - al * d.OW;
        const int n = n0 + ib, a = a0 + al;
        if (n < d.N && a < d.OH) v = d.dy[(long)n * d.dysn + (long)(cout0 + i) * d.dysc + a * d.OW + b];
      }
      ldsY[i * p.YS + pix] = v;
    }
    // ---- stage input patch: (G*32) x IB x Rin x Wst, logical (upsampled, zero padded) coordinates ----
    const int perch = p.IB * p.ImgS;
    const int nch = p.G * 32;
    for (int e = tid; e < nch * perch; e += nthr) {
      const int c = e / perch;
      int rem = e - c * perch;
      const int ib = rem / p.ImgS;
      rem -= ib * p.ImgS;
      const int ry = rem / p.Wst, rx = rem - ry * p.Wst;
      const int iy = a0 * d.stride - d.pad + ry, ix = rx - d.pad;
      const int n = n0 + ib, ci = cin0 + c;
      float v = 0.f;
      if (ci < d.Cin && n < d.N && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl)
        v = d.x[(long)n * d.xsn + (long)ci * d.xsc + (iy >> d.ups) * d.Win + (ix >> d.ups)];
      ldsX[c * p.XS + ib * p.ImgS + rem] = v;
    }
    __syncthreads();
    if (d.gb && blockIdx.x == 0 && tid < 32) {
      const float* row = ldsY + tid * p.YS;
      for (int pix = 0; pix < p.BP; ++pix) bsum += row[pix];
    }
    if (wave_active) {
      const float* arow = ldsY + j * p.YS + kh;
      const float* xrow = ldsX + (wave * 32 + j) * p.XS;
      for (int kp = 0; kp < p.BPp; kp += 2) {
        const float av = arow[kp];
        const float* xb = xrow + pixoff[kp + kh];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int ky = t / d.KW, kx = t - ky * d.KW;  // KW is uniform; strength-reduced by the compiler per t
          const float bv = xb[ky * p.Wst + kx];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }

  if (wave_active) {
    const int c = cin_w + j;
    if (c < d.Cin) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = cout0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (o < d.Cout) atomicAdd(d.gW + ((long)o * d.Cin + c) * T + t, d.scale * acc[t][r]);
        }
    }
  }
  if (d.gb && blockIdx.x == 0 && tid < 32 && cout0 + tid < d.Cout) atomicAdd(d.gb + cout0 + tid, d.scale * bsum);
}

static inline int odd_up(int v) { return v | 1; }

void launch_wgrad(const WgradDesc& d, hipStream_t s) {
  const int T = d.KH * d.KW;
  DBM_CHECK(T == 1 || T == 9 || T == 16, "wgrad: supported kernels are 1x1, 3x3, 4x4");
  WgradPlan p;
  p.d = d;
  const int tiles = (d.Cin + 31) / 32;
  const int groups = (tiles + 3) / 4;
  p.G = (tiles + groups - 1) / groups;
  // band selection: whole images if small, else row bands of one image; keep LDS <= ~96 KB
  const int Wst = (d.OW - 1) * d.stride + d.KW;
  const long budget = 24000;  // floats
  auto cost = [&](int IB, int R) {
    const int Rin = (R - 1) * d.stride + d.KH;
    return (long)p.G * 32 * odd_up(IB * Rin * Wst) + 32L * odd_up((IB * R * d.OW + 1) & ~1) + IB * R * d.OW + 2;
  };
  int IB = 1, R = d.OH;
  if (cost(1, d.OH) <= budget) {
    while (IB * 2 <= d.N && IB * 2 * d.OH * d.OW <= 256 && cost(IB * 2, d.OH) <= budget) IB *= 2;
  } else {
    while (R > 1 && cost(1, R) > budget) --R;
  }
  DBM_CHECK(cost(IB, R) <= 39000, "wgrad: band does not fit in LDS");
  p.IB = IB; p.R = R;
  p.nbr = (d.OH + R - 1) / R;
  p.BP = IB * R * d.OW;
  p.BPp = (p.BP + 1) & ~1;
  p.YS = odd_up(p.BPp);
  p.Rin = (R - 1) * d.stride + d.KH;
  p.Wst = Wst;
  p.ImgS = p.Rin * p.Wst;
  p.XS = odd_up(IB * p.ImgS);
  const int imgGroups = (d.N + IB - 1) / IB;
  p.nbands = imgGroups * p.nbr;
  const int coutTiles = (d.Cout + 31) / 32;
  int S = (1024 + groups * coutTiles - 1) / (groups * coutTiles);  // aim at ~1024 workgroups
  if (S > p.nbands) S = p.nbands;
  if (S < 1) S = 1;
  const size_t lds = sizeof(float) * ((size_t)32 * p.YS + p.BPp + (size_t)p.G * 32 * p.XS);
  dim3 grid(groups, coutTiles, S), block(64 * p.G);
#define DBM_WG(TT)                                                                                          \
  do {                                                                                                      \
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024));                                                               \
    hipLaunchKernelGGL(wgrad_kernel<TT>, grid, block, lds, s, p);                                           \
  } while (0)
  if (g_profiler.enabled)
    g_profiler.begin(s, 1, 2.0 * (double)d.N * d.OH * d.OW * d.Cout * d.Cin * T);
  if (T == 1) DBM_WG(1);
  else if (T == 9) DBM_WG(9);
  else DBM_WG(16);
#undef DBM_WG
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}
