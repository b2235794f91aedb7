// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces every L.Convolution2D call site with >= 32 input channels on the hot path
// (reference srgan_train.py:292-331, 467-503, 506-523 offset convs, 626-634) for forward,
// and -- with transposed/flipped packed weights -- the data-gradient of the same layers.
//
// Work decomposition (CDNA4): one workgroup of WAVES wavefronts owns a 32(out-channel) x 32(output-
// position) tile.  Output positions are the flattened (n, a, b) index, so a tile may straddle
// images; each lane keeps its own position's gather offsets (one 32-bit offset per tap).
// The K dimension (taps x input channels) is split WAVES-ways across the wavefronts by input
// channel; the partial 32x32 accumulators are reduced through LDS and the epilogue (bias,
// residual axpy's, LeakyReLU, gradient mask) is applied by all threads with 128-byte coalesced
// stores along the position axis.  WAVES is 4 when the layer has >= 1024 tiles (tail, discriminator,
// inference crops) and 8 / 16 for the 9x9 trunk at batch 64, whose 162 - 324 tiles would otherwise
// leave most SIMDs empty: a 16-wavefront workgroup puts 4 short MFMA chains on every SIMD of its CU.
//
// MFMA operand mapping (cdna_hip_programming.md section 3): A[i = lane&31][k = lane>>5] is the
// packed weight wp[t][ci + (lane>>5)][cout0 + (lane&31)] -> a 2 x 128-byte coalesced load;
// B[k = lane>>5][j = lane&31] is x[n_j][ci + (lane>>5)][tap-shifted position j] -> gathered
// straight from global/L2 (the taps re-read the same lines, so they hit the vector L1);
// D[i][j] has j = lane&31 (position) and i = (r&3) + 8*(r>>2) + 4*(lane>>5) (out channel).
// At the fp32 MFMA rate (64 cycles per instruction per SIMD) two dword loads per MFMA keep
// the L1 below half of its bandwidth, so no LDS staging of operands is needed; what matters is
// latency: the loop is unrolled over the T taps of one channel pair and the 2T loads of the NEXT
// pair are issued before the T MFMAs of the current one.
#include "dbm_internal.h"
#include <algorithm>
#include <array>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// NPB > 0: the layer has at most NPB channel pairs per wavefront (the 9x9 trunk with 8/16-way split-K): ALL gathered
// B operands of the wavefront's K slice are requested up front -- they are first-touch reads of activations another
// XCD just wrote (Infinity-Cache latency, > 1 us), paid once per kernel instead of once per channel pair -- and only
// the (L2-hot) weights are streamed with a one-pair prefetch.  NPB == 0: streaming loop, one pair of prefetch.
// ROW (3x3, stride 1, no resize only): the three taps of a kernel row are adjacent in memory, so they are fetched by
// ONE 12-byte load per lane instead of three dword loads -- the texture-addresser, not the MFMA pipe, is what the
// gathers saturate first.  The word before / after a row may belong to the neighbouring row, channel or image (or to
// the 128-byte guard every activation buffer carries); such taps are zeroed by the validity mask as before.
struct __attribute__((packed, aligned(4))) f32x3 { float x, y, z; };

// Split-K reduction over the WAVES wavefronts through LDS + epilogue (bias, residual axpy's, LeakyReLU, gradient mask,
// accumulate), shared by the kernel forms below.  ks > 1 (cross-workgroup split-K): the workgroup's partial tile goes to
// d.ks_part with agent-scope write-through stores; the workgroup that arrives LAST at the tile's counter adds the ks slices
// in index order -- the result does not depend on who is last -- and runs the epilogue.
// The epilogue's global operands of one thread (PASSES outputs): loaded by igemm_epilogue_load, consumed by igemm_epilogue.
template <int WAVES> struct IgemmEpiOps {
  static constexpr int PASSES = 32 / (2 * WAVES);
  float r1[PASSES], r2[PASSES], y[PASSES], m[PASSES], b[PASSES], s[PASSES];
};
template <int WAVES>
__device__ __forceinline__ void igemm_epilogue_load(const ConvDesc& d, IgemmEpiOps<WAVES>& o, int tid, bool pv, int n, int a, int b, int cout0,
                                                    int oy0, int ox0) {
  constexpr int ROWS_PER_PASS = 2 * WAVES, PASSES = 32 / ROWS_PER_PASS;
  const long pix = (long)(a * d.so + oy0) * d.OWp + (b * d.so + ox0);
  const int irow = tid >> 5;
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int i = irow + ROWS_PER_PASS * q;
    const int c = cout0 + i;
    o.r1[q] = o.r2[q] = o.y[q] = o.b[q] = 0.f;
    o.m[q] = o.s[q] = 1.f;
    if (!pv || c >= d.Cout) continue;
    const long co = (long)c * d.ysc + pix;
    if (d.bias) o.b[q] = d.bias[c];
    if (d.ch_scale) o.s[q] = d.ch_scale[c];
    if (d.r1 && c < d.r1_nch) o.r1[q] = d.r1[(long)n * d.r1sn + co];
    if (d.r2) o.r2[q] = d.r2[(long)n * d.r2sn + co];
    if (d.accumulate) o.y[q] = d.y[(long)n * d.ysn + co];
    if (d.mask && c >= d.mask_c0) o.m[q] = d.mask[(long)n * d.masksn + co];
  }
}

// pre: operands already requested by the caller (the two-tile form requests BOTH tiles' operands before the first tile's stores: a
// load issued behind a store is awaited by draining the store -- vmcnt is in order over both -- a write round trip per workgroup)
template <int WAVES>
__device__ __forceinline__ bool igemm_epilogue(const ConvDesc& d, float* red, const f32x16& acc, int tid, int wave, int j, int kh, bool pv,
                                               int n, int a, int b, int cout0, int ks, int kz, unsigned tile, int oy0, int ox0,
                                               const IgemmEpiOps<WAVES>* pre = nullptr) {
  const long pix = (long)(a * d.so + oy0) * d.OWp + (b * d.so + ox0);
  constexpr int ROWS_PER_PASS = 2 * WAVES;        // threads / 32
  constexpr int PASSES = 32 / ROWS_PER_PASS;      // 4, 2, 1 for WAVES = 4, 8, 16
  const int irow = tid >> 5;
  // Two phases: first every global read of the epilogue (residuals, accumulate target, mask) for ALL of this thread's
  // outputs is issued -- before the cross-wavefront reduction, whose LDS traffic and barrier hide their latency --
  // then the arithmetic and the stores.  (One pass at a time, the compiler cannot hoist the next
  // pass's loads above the previous pass's store -- y, r1, r2 and mask may alias -- and a short-K layer then spends
  // more time in PASSES serialised memory round trips than in its MFMAs.)
  IgemmEpiOps<WAVES> ops;
  float (&e_r1)[PASSES] = ops.r1;
  float (&e_r2)[PASSES] = ops.r2;
  float (&e_y)[PASSES] = ops.y;
  float (&e_m)[PASSES] = ops.m;
  float (&e_b)[PASSES] = ops.b;
  float (&e_s)[PASSES] = ops.s;
  auto load_operands = [&]() { igemm_epilogue_load<WAVES>(d, ops, tid, pv, n, a, b, cout0, oy0, ox0); };
  if (pre) ops = *pre;
  else if (ks <= 1) load_operands();
  // split-K reduction through LDS
  float* mine = red + wave * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * kh;
    mine[i * 32 + j] = acc[r];
  }
  __syncthreads();
  float vs[PASSES];
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int e = (irow + ROWS_PER_PASS * q) * 32 + j;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[w * 1024 + e];
    vs[q] = v;
  }
  if (ks > 1) {
    float* part = d.ks_part + ((size_t)tile * ks) * 1024;
#pragma unroll
    for (int q = 0; q < PASSES; ++q)
      __hip_atomic_store(part + (size_t)kz * 1024 + (irow + ROWS_PER_PASS * q) * 32 + j, vs[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wavefront's write-through stores have been acknowledged
    __syncthreads();                      // ... and so have the other wavefronts' (the LDS words below are free again, too)
    unsigned* flag = reinterpret_cast<unsigned*>(red);
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(d.ks_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != (unsigned)(ks - 1)) return false;
    if (tid == 0) __hip_atomic_store(d.ks_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    load_operands();
    // (the loads bypass the caches: a microsecond each -- eight slices of all passes are in flight together; slices past
    // the last one re-read it and add nothing)
#pragma unroll
    for (int q = 0; q < PASSES; ++q) vs[q] = 0.f;
    for (int z0 = 0; z0 < ks; z0 += 8) {
      float t[PASSES][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int z = z0 + u < ks ? z0 + u : ks - 1;
#pragma unroll
        for (int q = 0; q < PASSES; ++q)
          t[q][u] = __hip_atomic_load(part + (size_t)z * 1024 + (irow + ROWS_PER_PASS * q) * 32 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) vs[q] += z0 + u < ks ? t[q][u] : 0.f;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int i = irow + ROWS_PER_PASS * q;
    const int c = cout0 + i;
    if (!pv || c >= d.Cout) continue;
    const long co = (long)c * d.ysc + pix;
    float* yp = d.y + (long)n * d.ysn + co;
    float v = (vs[q] * e_s[q] + e_b[q]) * d.s1 + d.r1s * e_r1[q];
    if (d.r2) v = d.s2 * v + e_r2[q];
    v += e_y[q];
    if (d.act) v = v >= 0.f ? v : d.slope * v;
    v = e_m[q] >= 0.f ? v : d.slope * v;
    *yp = v;
  }
  return true;
}

// Epilogue of the no-split form (ConvDesc::nosplit): the wavefront owns its 32 x 32 tile over the whole K, so the
// accumulators go out straight from the registers -- no LDS, no barrier; lanes j = consecutive positions (128-byte runs).
__device__ __forceinline__ void igemm_epilogue_ns(const ConvDesc& d, const f32x16& acc, int kh, bool pv, int n, int a, int b, int cout0,
                                                  int oy0, int ox0) {
  if (!pv) return;
  const long pix = (long)(a * d.so + oy0) * d.OWp + (b * d.so + ox0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // four channels at a time: their global reads first (they overlap), then arithmetic and stores
    float e_r1[4], e_r2[4], e_y[4], e_m[4], e_b[4], e_s[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int c = cout0 + rr + 8 * q + 4 * kh;
      e_r1[rr] = e_r2[rr] = e_y[rr] = e_b[rr] = 0.f;
      e_m[rr] = e_s[rr] = 1.f;
      if (c >= d.Cout) continue;
      const long co = (long)c * d.ysc + pix;
      if (d.bias) e_b[rr] = d.bias[c];
      if (d.ch_scale) e_s[rr] = d.ch_scale[c];
      if (d.r1 && c < d.r1_nch) e_r1[rr] = d.r1[(long)n * d.r1sn + co];
      if (d.r2) e_r2[rr] = d.r2[(long)n * d.r2sn + co];
      if (d.accumulate) e_y[rr] = d.y[(long)n * d.ysn + co];
      if (d.mask && c >= d.mask_c0) e_m[rr] = d.mask[(long)n * d.masksn + co];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int c = cout0 + rr + 8 * q + 4 * kh;
      if (c >= d.Cout) continue;
      float v = (acc[4 * q + rr] * e_s[rr] + e_b[rr]) * d.s1 + d.r1s * e_r1[rr];
      if (d.r2) v = d.s2 * v + e_r2[rr];
      v += e_y[rr];
      if (d.act) v = v >= 0.f ? v : d.slope * v;
      v = e_m[rr] >= 0.f ? v : d.slope * v;
      d.y[(long)n * d.ysn + (long)c * d.ysc + pix] = v;
    }
  }
}

template <int T, int WAVES, int NPB, bool ROW, bool NS = false>
__global__ __launch_bounds__(64 * WAVES) void igemm_conv_kernel(const ConvDesc d) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // WAVES * 1024 floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int kh = lane >> 5;
  const int ks = d.ksplit > 1 ? d.ksplit : 1;
  // merged phases of a stride-2 data gradient (T == 4): blockIdx.z = phase * ks + K slice
  int OHl = d.OHl, OWl = d.OWl, oy0 = d.oy0, ox0 = d.ox0, tb = 0, kz = (int)blockIdx.z, ph = 0;
  unsigned planeM = d.planeM, owM = d.owM;
  const float* wp = d.wp;
  if constexpr (T == 4) {
    if (d.nphase > 1) {
      ph = (int)blockIdx.z / ks;
      kz = (int)blockIdx.z - ph * ks;
      OHl = d.phOH[ph]; OWl = d.phOW[ph]; planeM = d.phPlaneM[ph]; owM = d.phOwM[ph];
      wp = d.phwp[ph]; oy0 = ph >> 1; ox0 = ph & 1; tb = 4 * ph;
    }
  }
  const int plane = OHl * OWl;
  // no-split form (large grids): every wavefront owns a position tile of its own over the WHOLE K
  constexpr bool ns = NS;  // (its own instantiation: the register-resident epilogue must not cost the split forms their occupancy)
  const long ptile = ns ? (long)blockIdx.x * WAVES + wave : (long)blockIdx.x;
  if (ptile * 32 >= (long)d.N * plane) return;  // a phase with fewer positions than the launch's widest one / ragged last workgroup
  const long P = ptile * 32 + j;
  const bool pv = P < (long)d.N * plane;
  int n = 0, a = 0, b = 0;
  if (pv) {  // P < 2^31 (checked by the launcher): multiply-high estimates are at most one short
    unsigned q = __umulhi((unsigned)P, planeM);
    unsigned r = (unsigned)P - q * (unsigned)plane;
    if (r >= (unsigned)plane) { ++q; r -= (unsigned)plane; }
    n = (int)q;
    unsigned qa = __umulhi(r, owM);
    unsigned rb = r - qa * (unsigned)OWl;
    if (rb >= (unsigned)OWl) { ++qa; rb -= (unsigned)OWl; }
    a = (int)qa;
    b = (int)rb;
  }
  const int cout0 = blockIdx.y * (NPB == -4 ? 64 : 32);
  const unsigned tile = ((unsigned)ph * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;  // split-K: partial-tile slot
  const int cpw = ns ? d.Cin : d.Cin / ks / WAVES;                   // input channels per wavefront (even)
  const int c0 = ns ? kh : kz * (d.Cin / ks) + wave * cpw + kh;      // first input channel of this lane
  const int Hl = d.Hin << d.ups, Wl = d.Win << d.ups;
  const float* xn = d.x + (long)n * d.xsn + (long)c0 * d.xsc;
  const float* wlane = wp + (long)c0 * d.CoutP + cout0 + j;
  const long wtap = (long)d.Cin * d.CoutP;
  const long wstep = 2L * d.CoutP;
  const long xstep = 2L * d.xsc;
  const int npairs = cpw >> 1;

  // per-tap gather offsets; out-of-image taps read offset 0 (valid memory) and are zeroed by the mask
  int xoff[T];
  unsigned okmask = 0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int iy = a * d.sin + d.dy[tb + t];
    const int ix = b * d.sin + d.dx[tb + t];
    const bool ok = pv && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
    xoff[t] = ok ? (iy >> d.ups) * d.Win + (ix >> d.ups) : 0;
    okmask |= ok ? (1u << t) : 0u;
  }

  int roff[3] = {0, 0, 0};
  if constexpr (ROW) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = a + d.dy[3 * r];
      roff[r] = (pv && (unsigned)iy < (unsigned)Hl) ? iy * d.Win + b - 1 : 0;
    }
  }
  const bool rev = ROW && d.dx[0] > 0;  // data-gradient tap order: dx = +1, 0, -1
  auto load_b = [&](const float* xc, float (&out)[T]) {
    if constexpr (ROW) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f32x3 v = *reinterpret_cast<const f32x3*>(xc + roff[r]);
        out[3 * r + 0] = rev ? v.z : v.x;
        out[3 * r + 1] = v.y;
        out[3 * r + 2] = rev ? v.x : v.z;
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) out[t] = xc[xoff[t]];
    }
  };

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

  if constexpr (NPB > 0) {
    float bq[NPB][T];
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
      if (q < npairs) {
        load_b(xn + q * xstep, bq[q]);
      }
    }
    float av[T];
#pragma unroll
    for (int t = 0; t < T; ++t) av[t] = wlane[t * wtap];
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
      if (q < npairs) {
        float an[T];
        const float* wc = wlane + ((q + 1 < npairs) ? q + 1 : q) * wstep;
#pragma unroll
        for (int t = 0; t < T; ++t) an[t] = wc[t * wtap];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float bm = ((okmask >> t) & 1u) ? bq[q][t] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bm, acc, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) av[t] = an[t];
      }
    }
  } else if constexpr (NPB == -2) {
    // bf16 multiply, fp32 accumulate (DBM_BF16 inference): v_mfma_f32_32x32x16_bf16 consumes sixteen input channels
    // per instruction.  A lane's eight K values of the weight operand are ONE 16-byte load from the bf16 image; the
    // activations stay fp32 in memory (eight gathers per MFMA, as in the fp32 path) and are rounded to nearest-even
    // with v_cvt_pk_bf16_f32.  The matrix pipe is no longer the limit (32 instead of 512 cycles per sixteen
    // channels); the vector L1 is.  Groups of sixteen channels are dealt round-robin to the wavefronts.
    const int groups = d.Cin >> 4;
    const bf16x8* w16 = reinterpret_cast<const bf16x8*>(d.wp16);
    const float* xb = d.x + (long)n * d.xsn + (long)(8 * kh) * d.xsc;
    const bf16x8* wb = w16 + (long)kh * d.CoutP + cout0 + j;
    const long wt = (long)groups * 2 * d.CoutP;   // bf16x8 units between taps
    const long wgs = 2L * d.CoutP;                // ... between channel groups
    const long xgs = 16L * d.xsc;
    if constexpr (ROW) {
      // 3x3 unit-stride layers (the trunk of an inference crop): a kernel row's three taps are ONE 12-byte load per
      // channel, as in the fp32 row form -- 24 gathers + 9 weight loads per sixteen channels instead of 72 + 9 -- and a
      // whole channel group's requests are in flight before its nine MFMAs (the gathers, not the 32-cycle bf16 MFMAs,
      // bound this kernel: 63 -> ~30 us per trunk layer of a 288 x 288 crop).
      for (int g = ns ? 0 : wave; g < groups; g += ns ? 1 : WAVES) {
        const float* xc = xb + g * xgs;
        f32x3 bx[3][8];
        bf16x8 aw[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) bx[r][i] = *reinterpret_cast<const f32x3*>(xc + (long)i * d.xsc + roff[r]);
#pragma unroll
        for (int t = 0; t < 9; ++t) aw[t] = wb[g * wgs + t * wt];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int r = t / 3, k = t % 3;
          const bool ok = (okmask >> t) & 1u;
          bf16x8 bv;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f32x3& v = bx[r][i];
            const float e = k == 1 ? v.y : ((k == 0) != rev ? v.x : v.z);
            bv[i] = (__bf16)(ok ? e : 0.f);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[t], bv, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    // out-of-image taps read element 0 of the plane (always valid memory) and are zeroed after the load: no
    // conditional loads in the loop; one (group, tap) step of software pipelining
    float br[8], bn[8];
    bf16x8 an;
    const int gstep = ns ? 1 : WAVES;
    int g = ns ? 0 : wave;
    if (g < groups) {
      const float* xc = xb + g * xgs;
#pragma unroll
      for (int i = 0; i < 8; ++i) bn[i] = xc[(long)i * d.xsc + xoff[0]];
      an = wb[g * wgs];
    }
    for (; g < groups; g += gstep) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const bf16x8 av = an;
#pragma unroll
        for (int i = 0; i < 8; ++i) br[i] = bn[i];
        // request the next step: tap t + 1 of this group, or tap 0 of this wavefront's next group
        {
          const bool last = (t == T - 1);
          const int gn = last ? g + gstep : g;
          const int tn = last ? 0 : t + 1;
          if (gn < groups) {
            const float* xq = xb + gn * xgs + xoff[tn];
#pragma unroll
            for (int i = 0; i < 8; ++i) bn[i] = xq[(long)i * d.xsc];
            an = wb[gn * wgs + tn * wt];
          }
        }
        const bool ok = (okmask >> t) & 1u;
        bf16x8 bv;
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[i] = (__bf16)(ok ? br[i] : 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      }
    }
    }
  } else if constexpr (NPB == -4) {
    // TWO output-channel tiles per wavefront (layers with >= 64 output channels): the gathered B operand -- what the
    // texture addresser and the vector L1 saturate on first -- is loaded once and feeds two MFMA chains; only the
    // (coalesced, L2-hot) weight loads double.  blockIdx.y counts 64-channel groups; the second tile's accumulator is
    // `acc2` and goes through the same epilogue after the first.
    // Software pipeline: two register sets in ping-pong.  (A single set refilled by copies `cur = next` makes hipcc
    // interleave the copies with the MFMAs, and each copy waits for a load that has only just been issued: one exposed
    // memory latency per channel pair.  The sched_barriers keep the requests in front of the MFMA block they overlap.)
    float a0[T], w0[T], b0[T], a1[T], w1[T], b1[T];
    auto load_set = [&](int p, float (&a)[T], float (&w)[T], float (&b)[T]) {
      const float* wc = wlane + p * wstep;
#pragma unroll
      for (int t = 0; t < T; ++t) { a[t] = wc[t * wtap]; w[t] = wc[t * wtap + 32]; }
      load_b(xn + p * xstep, b);
    };
    auto mfma_set = [&](const float (&a)[T], const float (&w)[T], const float (&b)[T]) {
      __builtin_amdgcn_s_setprio(1);  // (a wavefront inside its MFMA block outranks the ones that are issuing loads)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float bm = ((okmask >> t) & 1u) ? b[t] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bm, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], bm, acc2, 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    };
    load_set(0, a0, w0, b0);
    for (int p = 0; p < npairs; p += 2) {
      load_set(p + 1 < npairs ? p + 1 : p, a1, w1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a0, w0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < npairs) {
        load_set(p + 2 < npairs ? p + 2 : p + 1, a0, w0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(a1, w1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if constexpr (NPB < 0) {
    // (NPB = -1) 4x4 stride-2 layers on tiny planes (2x2 -> 1x1: 12 of the 16 taps fall outside the image for EVERY position of
    // the tile): taps that no lane of the wavefront needs are skipped, loads and MFMA alike
    unsigned live = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) live |= (__ballot((okmask >> t) & 1u) != 0ull) ? (1u << t) : 0u;
    float av[T], bv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      av[t] = 0.f; bv[t] = 0.f;
      if (live & (1u << t)) { av[t] = wlane[t * wtap]; bv[t] = xn[xoff[t]]; }
    }
    for (int p = 0; p < npairs; ++p) {
      const int pn = (p + 1 < npairs) ? p + 1 : p;
      const float* xc = xn + pn * xstep;
      const float* wc = wlane + pn * wstep;
      float an[T], bn[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        an[t] = 0.f; bn[t] = 0.f;
        if (live & (1u << t)) { an[t] = wc[t * wtap]; bn[t] = xc[xoff[t]]; }
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (live & (1u << t)) {
          const float bm = ((okmask >> t) & 1u) ? bv[t] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bm, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        av[t] = an[t];
        bv[t] = bn[t];
      }
    }
  } else {
    // streaming loop, two register sets in ping-pong (see the two-tile form above)
    float a0[T], b0[T], a1[T], b1[T];
    auto load_set = [&](int p, float (&a)[T], float (&b)[T]) {
      const float* wc = wlane + p * wstep;
#pragma unroll
      for (int t = 0; t < T; ++t) a[t] = wc[t * wtap];
      load_b(xn + p * xstep, b);
    };
    auto mfma_set = [&](const float (&a)[T], const float (&b)[T]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float bm = ((okmask >> t) & 1u) ? b[t] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bm, acc, 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    };
    load_set(0, a0, b0);
    for (int p = 0; p < npairs; p += 2) {
      load_set(p + 1 < npairs ? p + 1 : p, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (p + 1 < npairs) {
        load_set(p + 2 < npairs ? p + 2 : p + 1, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if constexpr (NS) {
    igemm_epilogue_ns(d, acc, kh, pv, n, a, b, cout0, oy0, ox0);
    if constexpr (NPB == -4) igemm_epilogue_ns(d, acc2, kh, pv, n, a, b, cout0 + 32, oy0, ox0);
  } else if constexpr (NPB == -4) {  // two output tiles: partial-tile slots 2 tile, 2 tile + 1
    if (ks <= 1) {  // (both tiles' operands before the first tile's stores)
      IgemmEpiOps<WAVES> o1, o2;
      igemm_epilogue_load<WAVES>(d, o1, tid, pv, n, a, b, cout0, oy0, ox0);
      igemm_epilogue_load<WAVES>(d, o2, tid, pv, n, a, b, cout0 + 32, oy0, ox0);
      igemm_epilogue<WAVES>(d, red, acc, tid, wave, j, kh, pv, n, a, b, cout0, ks, kz, 2 * tile, oy0, ox0, &o1);
      __syncthreads();  // the first tile's partial sums have been read
      igemm_epilogue<WAVES>(d, red, acc2, tid, wave, j, kh, pv, n, a, b, cout0 + 32, ks, kz, 2 * tile + 1, oy0, ox0, &o2);
    } else {
      igemm_epilogue<WAVES>(d, red, acc, tid, wave, j, kh, pv, n, a, b, cout0, ks, kz, 2 * tile, oy0, ox0);
      __syncthreads();  // the first tile's partial sums (and the arrival word) have been read
      igemm_epilogue<WAVES>(d, red, acc2, tid, wave, j, kh, pv, n, a, b, cout0 + 32, ks, kz, 2 * tile + 1, oy0, ox0);
    }
  } else {
    igemm_epilogue<WAVES>(d, red, acc, tid, wave, j, kh, pv, n, a, b, cout0, ks, kz, tile, oy0, ox0);
  }
}


KernelProfiler g_profiler;

// ----------------------------------------------------------------------------------------------------------------------
// Position-major form for the deep discriminator layers (planes of <= 4 x 4 pixels: conv_layer6 .. 9 and the data gradients
// of the 3x3 ones; round 3, VERDICT next #4a).  On such planes most taps of most output positions fall into the padding --
// 3x3 on 2 x 2: 4 of 9 taps are inside the image for every position, 4x4 stride 2 on 4 x 4 -> 2 x 2: 9 of 16, 2 x 2 -> 1 x 1:
// 4 of 16, 3x3 on 4 x 4: 6.25 of 9 on average -- but WHICH taps differs from position to position, so the (n, a, b)-flattened
// tiles of the general form (eight images x four positions per tile) need every tap.  Here a tile is ONE output position x 32
// images (lane j = image): the set of live taps is the same for the whole tile, is found once (ballot over the taps, a compact
// {input offset, weight offset} list in LDS) and the K loop runs over live taps only -- loads and MFMAs of the padding are not
// issued at all.  Every step is one live tap x four channel pairs x two output-channel tiles (12 loads, 8 MFMAs), two register
// sets in ping-pong; offsets of a step are wavefront-uniform (SGPRs), so a lane holds no per-tap state.  Split-K across the four
// wavefronts and across workgroups, partial tiles and the epilogue exactly as in the general form (igemm_epilogue).
// ----------------------------------------------------------------------------------------------------------------------
template <int T, bool MT2>
__global__ __launch_bounds__(256) void igemm_pm_kernel(const ConvDesc d) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // 4 x 1024 floats
  constexpr int WAVES = 4, U = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int kh = lane >> 5;
  const int ks = d.ksplit > 1 ? d.ksplit : 1;
  const int kz = (int)blockIdx.z;
  const int pos = (int)blockIdx.x / d.pm_groups;
  const int grp = (int)blockIdx.x - pos * d.pm_groups;
  const int a = pos / d.OWl, b = pos - a * d.OWl;
  const int n = grp * 32 + j;
  const bool pv = n < d.N;
  const int cout0 = blockIdx.y * (MT2 ? 64 : 32);
  const unsigned tile = blockIdx.y * gridDim.x + blockIdx.x;
  const int cpw = d.Cin / ks / WAVES;                               // input channels per wavefront: a multiple of 8
  // buffer loads: descriptor (kernel-uniform) + loop-invariant 32-bit lane offset + a wavefront-uniform SGPR offset per load --
  // the K loop has no VGPR address arithmetic (with 64-bit flat addresses hipcc recycles the loaded registers of one set as
  // address temporaries of the other and then waits for the loads in flight before every address it forms)
  const int cw = kz * (d.Cin / ks) + __builtin_amdgcn_readfirstlane(wave) * cpw;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wp), 0, 0x7fffffff, 0x00020000);
  const int xl = 4 * ((pv ? n : 0) * (int)d.xsn + kh * d.xsc);  // (lanes past the batch read image 0: their columns are dropped)
  const int wl = 4 * (kh * d.CoutP + j);
  const int xu = 4 * cw * d.xsc;                   // uniform byte offsets of the wavefront's first channel
  const int wu = 4 * (cw * d.CoutP + cout0);
  const int wtap = 4 * d.Cin * d.CoutP;            // bytes between taps / channel pairs
  const int wstep = 8 * d.CoutP;
  const int xstep = 8 * d.xsc;
  // the live taps of this output position: tap t's input offset and weight offset are computed by lane t and pushed to lane
  // rank(t) (ds_permute: a compact list across the lanes of a register); step s of the K loop fetches the offsets of live tap
  // s >> gshift with v_readlane -- wavefront-uniform SGPRs, no LDS, no per-lane tap state, no branch in the loop
  bool ok = false;
  int xo = 0;
  if (lane < T) {
    const int iy = a * d.sin + d.dy[lane], ix = b * d.sin + d.dx[lane];
    ok = (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
    xo = iy * d.Win + ix;
  }
  const unsigned live = (unsigned)__ballot(ok);
  const int dst = (ok ? __popc(live & ((1u << (lane & 31)) - 1u)) : 32 + j) << 2;   // (dead lanes push into the unused upper half)
  const int xs = __builtin_amdgcn_ds_permute(dst, 4 * xo);       // (bytes)
  const int ws = __builtin_amdgcn_ds_permute(dst, lane * wtap);   // (bytes)
  const int S = __popc(live) << d.pm_gshift;                         // steps: live taps x sets of U channel pairs
  const int gmask = (1 << d.pm_gshift) - 1;

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  float a0[U], w0[U], b0[U], a1[U], w1[U], b1[U];
  auto load_set = [&](int s, float (&av)[U], float (&wv)[U], float (&bv)[U]) {
    const int i = s >> d.pm_gshift;
    const int q = (s & gmask) * U;
    const int wc = wu + __builtin_amdgcn_readlane(ws, i) + q * wstep;
    const int xc = xu + __builtin_amdgcn_readlane(xs, i) + q * xstep;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      av[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wl, wc + u * wstep, 0));
      if constexpr (MT2) wv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wl, wc + u * wstep + 128, 0));
      bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xl, xc + u * xstep, 0));
    }
  };
  auto mfma_set = [&](const float (&av)[U], const float (&wv)[U], const float (&bv)[U]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      if constexpr (MT2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u], bv[u], acc2, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if (S > 0) {
    load_set(0, a0, w0, b0);
    // (both exits leave the loop directly: with a shared latch hipcc's wait-count pass has to assume that the other set's loads
    // are still in flight at the loop head and waits for this set's before it may reuse an address register)
    for (int s = 0;;) {
      load_set(s + 1 < S ? s + 1 : s, a1, w1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a0, w0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 >= S) break;
      load_set(s + 2 < S ? s + 2 : s + 1, a0, w0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a1, w1, b1);
      __builtin_amdgcn_sched_barrier(0);
      s += 2;
      if (s >= S) break;
    }
  }
  if constexpr (MT2) {
    igemm_epilogue<WAVES>(d, red, acc, tid, wave, j, kh, pv, n, a, b, cout0, ks, kz, 2 * tile, d.oy0, d.ox0);
    __syncthreads();
    igemm_epilogue<WAVES>(d, red, acc2, tid, wave, j, kh, pv, n, a, b, cout0 + 32, ks, kz, 2 * tile + 1, d.oy0, d.ox0);
  } else {
    igemm_epilogue<WAVES>(d, red, acc, tid, wave, j, kh, pv, n, a, b, cout0, ks, kz, tile, d.oy0, d.ox0);
  }
}

void KernelProfiler::begin(hipStream_t s, int family, double flops, double bytes, const char* tag, long wgs) {
  if (serial) DBM_HIP(hipDeviceSynchronize());  // standalone durations: nothing else is running when the bracket opens
  Rec r;
  DBM_HIP(hipEventCreate(&r.a));
  DBM_HIP(hipEventCreate(&r.b));
  r.flops = flops;
  r.bytes = bytes;
  r.family = family;
  r.wgs = wgs;
  snprintf(r.tag, sizeof(r.tag), "%s", tag ? tag : "-");
  DBM_HIP(hipEventRecord(r.a, s));
  recs.push_back(r);
}

void KernelProfiler::end(hipStream_t s) {
  DBM_HIP(hipEventRecord(recs.back().b, s));
  if (serial) DBM_HIP(hipDeviceSynchronize());  // ... and nothing else starts before it closes
}

void KernelProfiler::collect(double* out, int nfam) {
  for (int i = 0; i < 3 * nfam; ++i) out[i] = 0.0;
  for (auto& r : recs) {
    DBM_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    DBM_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    const int f = (r.family == 4 && nfam <= 4) ? 2 : r.family;  // (callers that ask for four families: both trunk forms)
    if (f < nfam) {
      out[f * 3 + 0] += ms;
      out[f * 3 + 1] += r.flops;
      out[f * 3 + 2] += 1.0;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  recs.clear();
}

std::string KernelProfiler::dump_records() {
  std::string out;
  for (auto& r : recs) {
    DBM_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    DBM_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    char line[200];
    snprintf(line, sizeof(line), "%d %.6e %.6e %.6f %ld %s\n", r.family, r.flops, r.bytes, ms, r.wgs, r.tag);
    out += line;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  recs.clear();
  return out;
}

void KernelProfiler::mark(hipStream_t s, const char* name) {
  hipEvent_t e;
  DBM_HIP(hipEventCreate(&e));
  DBM_HIP(hipEventRecord(e, s));
  marks.emplace_back(name, e);
}

std::string KernelProfiler::dump_marks() {
  std::string out;
  for (auto& m : marks) {
    DBM_HIP(hipEventSynchronize(m.second));
    float ms = 0.f;
    DBM_HIP(hipEventElapsedTime(&ms, marks.front().second, m.second));
    char line[160];
    snprintf(line, sizeof(line), "%s %.4f\n", m.first.c_str(), ms);
    out += line;
  }
  for (auto& m : marks) (void)hipEventDestroy(m.second);
  marks.clear();
  return out;
}

static bool igemm_tap_skip(const ConvDesc& d) {
  static const int skip_plane = DBM_TUNE_GETENV("IGEMM_SKIP_PLANE") ? atoi(DBM_TUNE_GETENV("IGEMM_SKIP_PLANE")) : 4;
  return (d.T == 16 || d.T == 4) && d.Hin * d.Win <= skip_plane;
}

constexpr int IGEMM_NPB = 6;  // channel pairs a wavefront may hold entirely in registers (T = 9 -> 54 VGPRs)



template <int T, int WAVES, bool ROW>
static void launch_twr(const ConvDesc& d, dim3 grid, hipStream_t s, bool mt2) {
  if (d.wp16) {
    if constexpr (!ROW || T == 9) {
      if constexpr (WAVES == 4) {
        if (d.nosplit) {
          hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, -2, ROW, true>), grid, dim3(64 * WAVES), 0, s, d);
          return;
        }
      }
      hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, -2, ROW>), grid, dim3(64 * WAVES), WAVES * 4096, s, d);
      return;
    }
  }
  if constexpr (WAVES == 4) {
    if (d.nosplit) {  // (decided by the launcher: large grids, no cross-workgroup split)
      if (mt2) hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, -4, ROW, true>), grid, dim3(64 * WAVES), 0, s, d);
      else hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, 0, ROW, true>), grid, dim3(64 * WAVES), 0, s, d);
      return;
    }
    if (mt2) {  // two output tiles per wavefront: grid.y counts 64-channel groups
      hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, -4, ROW>), grid, dim3(64 * WAVES), WAVES * 4096, s, d);
      return;
    }
  }
  const int npairs = d.Cin / (d.ksplit > 1 ? d.ksplit : 1) / WAVES / 2;
  if constexpr (T <= 9 && WAVES >= 8) {
    if (npairs <= IGEMM_NPB) {
      hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, IGEMM_NPB, ROW>), grid, dim3(64 * WAVES), WAVES * 4096, s, d);
      return;
    }
  }
  if constexpr ((T == 16 || T == 4) && !ROW) {
    // most taps of a 4x4 window (of a 2x2 phase window of its data gradient) fall outside such planes: the tap-skipping variant
    if (igemm_tap_skip(d)) {
      hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, -1, ROW>), grid, dim3(64 * WAVES), WAVES * 4096, s, d);
      return;
    }
  }
  hipLaunchKernelGGL((igemm_conv_kernel<T, WAVES, 0, ROW>), grid, dim3(64 * WAVES), WAVES * 4096, s, d);
}

template <int T, int WAVES>
static void launch_tw(const ConvDesc& d, dim3 grid, hipStream_t s, bool mt2) {
  if constexpr (T == 9) {
    // row-contiguous taps: 3x3, unit stride, no folded resize, taps ordered (dy, dx) with dx = -1,0,1 or 1,0,-1
    const bool row = d.sin == 1 && d.ups == 0 && d.dy[0] == d.dy[1] && d.dy[1] == d.dy[2] && d.dx[1] == 0 &&
                     d.dx[0] == -d.dx[2] && (d.dx[0] == 1 || d.dx[0] == -1) && d.dy[3] == d.dy[5] && d.dy[6] == d.dy[8] &&
                     d.dx[3] == d.dx[0] && d.dx[6] == d.dx[0] && d.dx[4] == 0 && d.dx[7] == 0;
    static const int bf16_row = DBM_TUNE_GETENV("BF16_ROW") ? atoi(DBM_TUNE_GETENV("BF16_ROW")) : 1;
    if (row && (!d.wp16 || bf16_row)) {
      launch_twr<T, WAVES, true>(d, grid, s, mt2);
      return;
    }
  }
  launch_twr<T, WAVES, false>(d, grid, s, mt2);
}

template <int T>
static void launch_t(const ConvDesc& d, dim3 grid, int waves, hipStream_t s, bool mt2) {
  if (waves == 16) launch_tw<T, 16>(d, grid, s, false);
  else if (waves == 8) launch_tw<T, 8>(d, grid, s, false);
  else launch_tw<T, 4>(d, grid, s, mt2);
}

// Workspace of the cross-workgroup split-K (partial tiles + arrival counters), one per launch stream: launches on one
// stream never overlap, launches on different streams (the real- and the fake-batch passes of the discriminator) do.
struct KsWorkspace { float* part = nullptr; unsigned* cnt = nullptr; };
static const size_t KS_PART_FLOATS = 4u << 20;  // 16 MB: 4096 partial tiles
static const size_t KS_COUNTERS = 4096;
static KsWorkspace& ks_workspace(hipStream_t s) {
  static std::map<hipStream_t, KsWorkspace> table;
  KsWorkspace& w = table[s];
  if (!w.part) {
    DBM_HIP(hipMalloc((void**)&w.part, KS_PART_FLOATS * sizeof(float)));
    DBM_HIP(hipMalloc((void**)&w.cnt, KS_COUNTERS * sizeof(unsigned)));
    DBM_HIP(hipMemset(w.cnt, 0, KS_COUNTERS * sizeof(unsigned)));
    DBM_HIP(hipDeviceSynchronize());  // the NULL-stream memset vs. non-blocking streams
  }
  return w;
}

static unsigned igemm_magic(unsigned long long dv) {  // floor(2^32 / dv), saturated (dv == 1: the kernel's one-step correction still lands)
  const unsigned long long m = 0x100000000ULL / dv;
  return (unsigned)(m > 0xffffffffULL ? 0xffffffffULL : m);
}

// Per-shape launch configuration (measured inside the training step, where the neighbours decide what a workgroup count costs):
// key = (taps, Cin, Cout, positions of the widest phase, phases) -> two tiles per wavefront / wavefronts per tile / K split;
// -1 = what the rules below say.  DBM_IGEMM_OVERRIDE="T:Cin:Cout:positions:phases=mt2,waves,ks;..." adds entries (tuning aid),
// DBM_IGEMM_LOG=1 prints every distinct launch once.
struct IgemmForce { int mt2, waves, ks; };
typedef std::array<long, 5> IgemmKey;
static const std::map<IgemmKey, IgemmForce>& igemm_overrides() {
  static std::map<IgemmKey, IgemmForce> tab = [] {
    std::map<IgemmKey, IgemmForce> t;
    if (const char* e = DBM_TUNE_GETENV("IGEMM_OVERRIDE")) {
      std::string str(e);
      size_t pos = 0;
      while (pos < str.size()) {
        size_t end = str.find(';', pos);
        if (end == std::string::npos) end = str.size();
        long k[5]; int f[3];
        if (sscanf(str.substr(pos, end - pos).c_str(), "%ld:%ld:%ld:%ld:%ld=%d,%d,%d", &k[0], &k[1], &k[2], &k[3], &k[4], &f[0], &f[1], &f[2]) == 8)
          t[IgemmKey{k[0], k[1], k[2], k[3], k[4]}] = IgemmForce{f[0], f[1], f[2]};
        pos = end + 1;
      }
    }
    return t;
  }();
  return tab;
}

void launch_igemm_conv(const ConvDesc& d_in, hipStream_t s) {
  if (dbm_abl_skip() & 512) return;  // (libdbm_measure.so only)
  ConvDesc d = d_in;
  d.ksplit = 1;
  const int nph = d.nphase > 1 ? d.nphase : 1;
  long total = (long)d.N * d.OHl * d.OWl;   // positions of the (widest) phase
  double flop_positions = (double)total;    // ... of all phases
  if (nph > 1) {
    DBM_CHECK(d.T == 4 && nph == 4, "igemm: merged phases are the four 2x2-tap phases of a k4 s2 data gradient");
    total = 0; flop_positions = 0;
    for (int ph = 0; ph < 4; ++ph) {
      const long tp = (long)d.N * d.phOH[ph] * d.phOW[ph];
      DBM_CHECK(tp > 0, "igemm: empty phase");
      total = std::max(total, tp);
      flop_positions += (double)tp;
      d.phPlaneM[ph] = igemm_magic((unsigned long long)d.phOH[ph] * d.phOW[ph]);
      d.phOwM[ph] = igemm_magic((unsigned long long)d.phOW[ph]);
    }
  }
  DBM_CHECK(total < (1L << 31), "igemm: more than 2^31 output positions");
  d.planeM = igemm_magic((unsigned long long)d.OHl * d.OWl);
  d.owM = igemm_magic((unsigned long long)d.OWl);
  DBM_CHECK(d.Cin % 32 == 0, "igemm: Cin must be a multiple of 32");
  DBM_CHECK(d.CoutP % 32 == 0 && d.Cout <= d.CoutP, "igemm: bad CoutP");
  DBM_CHECK(d.T == 1 || d.T == 4 || d.T == 9 || d.T == 16, "igemm: tap count must be 1, 4, 9 or 16");
  {  // the deep discriminator layers (planes of <= 4 x 4): position-major tiles, live taps only (igemm_pm_kernel)
    const int pm_enable = DBM_TUNE_GETENV("IGEMM_PM") ? atoi(DBM_TUNE_GETENV("IGEMM_PM")) : 1;          // (read per call: A/B in one process)
    const int pm_target = DBM_TUNE_GETENV("IGEMM_PM_KSTARGET") ? atoi(DBM_TUNE_GETENV("IGEMM_PM_KSTARGET")) : 512;
    const int pm_min_n = DBM_TUNE_GETENV("IGEMM_PM_MIN_N") ? atoi(DBM_TUNE_GETENV("IGEMM_PM_MIN_N")) : 16;
    // (3x3 on 4 x 4 planes -- 6.25 of 9 taps live on average -- stays with the general form: measured 35.5 against 30.7 us;
    //  DBM_IGEMM_PM_K3_PLANE: largest plane of a 3x3 layer that takes this form)
    const int pm_k3_plane = DBM_TUNE_GETENV("IGEMM_PM_K3_PLANE") ? atoi(DBM_TUNE_GETENV("IGEMM_PM_K3_PLANE")) : 4;
    if (pm_enable && !d.wp16 && d.ups == 0 && nph == 1 && (d.T == 9 || d.T == 16) && d.Hin * d.Win <= (d.T == 9 ? pm_k3_plane : 16) && d.OHl * d.OWl <= 16 &&
        d.N >= pm_min_n && d.Cin % 32 == 0 && 4L * d.T * d.Cin * d.CoutP < (1L << 31) && 4L * (d.N + 32) * d.xsn < (1L << 31)) {  // (32-bit byte offsets)
      d.pm_groups = (d.N + 31) / 32;
      const bool mt2 = d.CoutP % 64 == 0 && d.Cout > 32;
      dim3 grid((unsigned)(d.OHl * d.OWl * d.pm_groups), (unsigned)((d.Cout + (mt2 ? 63 : 31)) / (mt2 ? 64 : 32)), 1u);
      const long tiles = (long)grid.x * grid.y;
      int ks = 1;  // input channels per workgroup stay a multiple of 32 (eight per wavefront)
      while (ks < 32 && tiles * ks * 2 <= pm_target && (d.Cin / (ks * 2)) % 32 == 0) ks *= 2;
      const size_t slots = (size_t)tiles * (mt2 ? 2 : 1);
      if (ks > 1 && !(slots * ks * 1024 <= KS_PART_FLOATS && slots <= KS_COUNTERS)) ks = 1;
      if (ks > 1) {
        KsWorkspace& w = ks_workspace(s);
        d.ksplit = ks;
        d.ks_part = w.part;
        d.ks_cnt = w.cnt;
        grid.z = (unsigned)ks;
      }
      const int sets = d.Cin / ks / 4 / 8;  // sets of four channel pairs per live tap and wavefront: a power of two?
      int sh = 0;
      while ((1 << sh) < sets) ++sh;
      if ((1 << sh) == sets && sets >= 1 && (d.Cin / ks) % 32 == 0) {
        d.pm_gshift = sh;
        if (g_profiler.enabled) {
          const double bytes = 4.0 * ((double)d.N * d.Cin * d.Hin * d.Win + flop_positions * d.Cout + (double)d.T * d.Cin * d.CoutP) +
                               (d.accumulate ? 4.0 * flop_positions * d.Cout : 0.0) + (d.mask ? 4.0 * flop_positions * (d.Cout - d.mask_c0) : 0.0);
          char tag[40];
          snprintf(tag, sizeof(tag), "c%d>%d_k%d_%dx%d_pm", d.Cin, d.Cout, d.T, d.Hin, d.Win);
          g_profiler.begin(s, 0, 2.0 * flop_positions * d.Cout * d.Cin * d.T, bytes, tag, (long)grid.x * grid.y * grid.z);
        }
        const size_t lds = 4 * 4096;
        if (d.T == 9) {
          if (mt2) hipLaunchKernelGGL((igemm_pm_kernel<9, true>), grid, dim3(256), lds, s, d);
          else hipLaunchKernelGGL((igemm_pm_kernel<9, false>), grid, dim3(256), lds, s, d);
        } else {
          if (mt2) hipLaunchKernelGGL((igemm_pm_kernel<16, true>), grid, dim3(256), lds, s, d);
          else hipLaunchKernelGGL((igemm_pm_kernel<16, false>), grid, dim3(256), lds, s, d);
        }
        if (g_profiler.enabled) g_profiler.end(s);
        DBM_HIP(hipGetLastError());
        return;
      }
      d.ksplit = 1; d.ks_part = nullptr; d.ks_cnt = nullptr; d.pm_groups = 0;  // (odd channel counts: the general form)
    }
  }
  {  // mid-size planes of the training step (18 x 18, 36 x 36 outputs): operands staged in LDS, no split-K (conv_tile.hip)
    long wgs = 0;
    const int cfg = conv_tile_plan(d, &wgs);
    if (cfg) {
      if (g_profiler.enabled) {
        double bytes = 4.0 * ((double)d.N * d.Cin * d.Hin * d.Win + flop_positions * d.Cout) + 4.0 * d.T * (double)d.Cin * d.CoutP;
        if (d.r1) bytes += 4.0 * flop_positions * d.r1_nch;
        if (d.r2) bytes += 4.0 * flop_positions * d.Cout;
        if (d.mask) bytes += 4.0 * flop_positions * (d.Cout - d.mask_c0);
        if (d.accumulate) bytes += 4.0 * flop_positions * d.Cout;
        char tag[40];
        snprintf(tag, sizeof(tag), "c%d>%d_k%d_%dx%d%s", d.Cin, d.Cout, d.T, d.Hin, d.Win, d.ups ? "u" : "");
        g_profiler.begin(s, 0, 2.0 * flop_positions * d.Cout * d.Cin * d.T, bytes, tag, wgs);
      }
      conv_tile_launch(d, cfg, s);
      if (g_profiler.enabled) g_profiler.end(s);
      return;
    }
  }
  dim3 grid((unsigned)((total + 31) / 32), (unsigned)((d.Cout + 31) / 32), (unsigned)nph);
  long tiles = (long)grid.x * grid.y * nph;
  // Two output-channel tiles per wavefront (the gathered B operand feeds two MFMA chains): layers with >= 64 output
  // channels on large grids -- or, with the cross-workgroup split-K below restoring the workgroup count, any layer
  // with a long K (DBM_IGEMM_MT2: 0 never, 1 large grids only, 2 also with split-K).
  static const int mt2_mode = DBM_TUNE_GETENV("IGEMM_MT2") ? atoi(DBM_TUNE_GETENV("IGEMM_MT2")) : 2;
  static const int ks_enable = DBM_TUNE_GETENV("IGEMM_KSPLIT") ? atoi(DBM_TUNE_GETENV("IGEMM_KSPLIT")) : 1;
  static const int ks_target = DBM_TUNE_GETENV("IGEMM_KSTARGET") ? atoi(DBM_TUNE_GETENV("IGEMM_KSTARGET")) : 256;
  // (round 3: 1024 -> 256 workgroups per split launch, 512 for the position-major form: inside the step these launches live on
  // the 64 CUs a persistent trunk launch leaves, where the number of workgroups, not the length of a K slice, is what they
  // pay for -- 8.22-8.30 ms per step with 1024 / the general form only, 8.13-8.16 with 256 / 512 and the position-major form)
  // (2048: conv_layer2 of the discriminator -- 1296 two-tile workgroups = 5.06 per CU, a sixth round on sixteen CUs -- stays
  // on one tile per wavefront, the 36 x 36 generator layers (2592) take two: 8.56 -> 8.47 ms per step against 1024)
  static const int mt2_tiles = DBM_TUNE_GETENV("IGEMM_MT2_TILES") ? atoi(DBM_TUNE_GETENV("IGEMM_MT2_TILES")) : 2048;
  const bool mt2_ok = mt2_mode && !d.wp16 && d.CoutP % 64 == 0 && grid.y % 2 == 0 && !igemm_tap_skip(d);
  bool mt2 = mt2_ok && tiles / 2 >= mt2_tiles;
  if (!mt2 && mt2_ok && mt2_mode >= 2 && ks_enable && (long)d.Cin * d.T >= 1024 && tiles / 2 <= 512 && tiles >= 64) mt2 = true;
  const IgemmKey key{d.T, d.Cin, d.Cout, total, nph};
  IgemmForce force{-1, -1, -1};
  if (!d.wp16) {
    auto it = igemm_overrides().find(key);
    if (it != igemm_overrides().end()) force = it->second;
  }
  if (force.mt2 >= 0) mt2 = force.mt2 && mt2_ok;
  if (mt2) { grid.y /= 2; tiles /= 2; }
  // few tiles -> more wavefronts per tile (Cin % 32 == 0 keeps Cin / WAVES even for every choice)
  // (1536 -- eight wavefronts for the 1296-tile layers, 5.06 four-wavefront workgroups per CU -- measured -0.04 ms per step; not
  // taken: the other summation order moves one discriminator gradient of the batch-64 fixture past its bound, a slope flip)
  static const int w4_tiles = DBM_TUNE_GETENV("IGEMM_W4_TILES") ? atoi(DBM_TUNE_GETENV("IGEMM_W4_TILES")) : 1024;
  static const int w8_tiles = DBM_TUNE_GETENV("IGEMM_W8_TILES") ? atoi(DBM_TUNE_GETENV("IGEMM_W8_TILES")) : 512;
  int waves = (tiles >= w4_tiles || mt2) ? 4 : (tiles >= w8_tiles ? 8 : 16);
  // ... but a wavefront should own a few channel pairs: with a short K (the 32-channel data gradients of the dense
  // blocks) the cross-wavefront reduction and a 1024-thread workgroup cost more than the MFMAs they spread
  static const int min_pairs = DBM_TUNE_GETENV("IGEMM_MINPAIRS") ? atoi(DBM_TUNE_GETENV("IGEMM_MINPAIRS")) : 4;  // (re-measured at the end of round 2: 6 -> 4, -0.08 ms per step)
  static const int min_tiles = DBM_TUNE_GETENV("IGEMM_MINTILES") ? atoi(DBM_TUNE_GETENV("IGEMM_MINTILES")) : 96;
  while (tiles > min_tiles && waves > 4 && d.Cin / (2 * waves) < min_pairs) waves >>= 1;
  if (force.waves > 0 && !(mt2 && force.waves != 4)) waves = force.waves;
  // Few tiles and a long K (the deep discriminator layers: 32..512 tiles, K = 2048..8192): the input channels are also split
  // across workgroups of four wavefronts, about 1024 workgroups per launch; partial tiles are folded deterministically by
  // the last workgroup of each tile (igemm_epilogue).  The bf16 inference images keep the one-workgroup form.
  if (ks_enable && (force.ks > 1 || (force.ks < 0 && tiles <= 512 && (long)d.Cin * d.T >= 1024)) && !d.wp16) {
    int ks = 1;
    while (ks < 32 && tiles * ks * 2 <= ks_target && (d.Cin / (ks * 2)) % 8 == 0 && d.Cin / (ks * 2) >= 32) ks *= 2;
    if (force.ks > 1 && (d.Cin / force.ks) % 8 == 0 && d.Cin / force.ks >= 32) ks = force.ks;
    const size_t slots = (size_t)tiles * (mt2 ? 2 : 1);
    if (ks > 1 && slots * ks * 1024 <= KS_PART_FLOATS && slots <= KS_COUNTERS) {
      KsWorkspace& w = ks_workspace(s);
      d.ksplit = ks;
      d.ks_part = w.part;
      d.ks_cnt = w.cnt;
      grid.z = (unsigned)(nph * ks);
      waves = 4;
    }
  }
  if (mt2 && waves != 4) { mt2 = false; grid.y *= 2; }
  {
    static const bool log = DBM_TUNE_GETENV("IGEMM_LOG") != nullptr;
    if (log) {
      static std::map<IgemmKey, int> seen;
      if (!seen.count(key)) {
        seen[key] = 1;
        fprintf(stderr, "igemm %ld:%ld:%ld:%ld:%ld tiles=%ld mt2_ok=%d -> mt2=%d waves=%d ks=%d\n", key[0], key[1], key[2], key[3], key[4],
                (long)grid.x * grid.y * nph, (int)mt2_ok, (int)mt2, waves, d.ksplit);
      }
    }
  }
  // bf16 inference on large grids (>= 2048 position tiles: the crops of the area sweep): no split-K at all -- each of a
  // workgroup's four wavefronts owns a position tile of its own over the whole K; no LDS reduction, no barrier, the
  // accumulators go out from the registers.  (With sixteen channels per 32-cycle MFMA a K slice is a handful of
  // instructions and the cross-wavefront reduction costs more than it spreads: 14.2 -> 13.2 ms per 288 x 288 crop.  The
  // fp32 layers keep the split: 88 vs 93-100 us on the 36 x 36 layers, 22.0 vs 25.4 ms per fp32 crop -- DBM_IGEMM_NOSPLIT_F32=1.)
  static const int ns_tiles = DBM_TUNE_GETENV("IGEMM_NOSPLIT") ? atoi(DBM_TUNE_GETENV("IGEMM_NOSPLIT")) : 2048;
  static const int ns_f32 = DBM_TUNE_GETENV("IGEMM_NOSPLIT_F32") ? atoi(DBM_TUNE_GETENV("IGEMM_NOSPLIT_F32")) : 0;
  d.nosplit = 0;
  if (ns_tiles > 0 && (d.wp16 || ns_f32) && d.ksplit <= 1 && waves == 4 && (long)grid.x >= ns_tiles) {
    d.nosplit = 1;
    grid.x = (grid.x + 3) / 4;
  }
  if (g_profiler.enabled) {
    // algorithmic bytes: the input once, the weight image(s) once, the output once (+ whatever the epilogue reads)
    const double np = d.nphase > 1 ? d.nphase : 1;
    double bytes = 4.0 * ((double)d.N * d.Cin * d.Hin * d.Win + flop_positions * d.Cout) + (d.wp16 ? 2.0 : 4.0) * np * d.T * (double)d.Cin * d.CoutP;
    if (d.r1) bytes += 4.0 * flop_positions * d.r1_nch;
    if (d.r2) bytes += 4.0 * flop_positions * d.Cout;
    if (d.mask) bytes += 4.0 * flop_positions * (d.Cout - d.mask_c0);
    if (d.accumulate) bytes += 4.0 * flop_positions * d.Cout;
    char tag[40];
    snprintf(tag, sizeof(tag), "c%d>%d_k%d%s_%dx%d%s%s", d.Cin, d.Cout, d.T, d.nphase > 1 ? "p" : "", d.Hin, d.Win, d.ups ? "u" : "", d.wp16 ? "_bf16" : "");
    g_profiler.begin(s, 0, 2.0 * flop_positions * d.Cout * d.Cin * d.T, bytes, tag, (long)grid.x * grid.y * grid.z);
  }
  switch (d.T) {
    case 1: launch_t<1>(d, grid, waves, s, mt2); break;
    case 4: launch_t<4>(d, grid, waves, s, mt2); break;
    case 9: launch_t<9>(d, grid, waves, s, mt2); break;
    default: launch_t<16>(d, grid, waves, s, mt2); break;
  }
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// weight packing
// ----------------------------------------------------------------------------------------------
// One workgroup per 32 (out) x TC (in) tile of a job, all taps: the OIHW rows are read in contiguous runs of TC * KH * KW
// floats, transposed through LDS and written in runs along the packed image's fastest axis (the element-wise gather this
// replaces read every float from a different cache line: discriminator repack 278 -> ~60 us).
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[32 * (32 * 9 + 1) + 1];   // (+ 1: the slot that absorbs the batched loop's out-of-range elements)
  int lo = 0, hi = njobs - 1;
  const int blk = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= blk) lo = mid; else hi = mid - 1;
  }
  const PackJob& p = jobs[lo];
  const int taps = p.KH * p.KW;
  const int TC = taps > 9 ? 16 : 32;             // input channels per tile (LDS: 32 x TC x taps floats)
  const int oR = p.transpose ? p.KP : p.MP;       // padded extents along the out / in axes of the OIHW tensor
  const int cR = p.transpose ? p.MP : p.KP;
  const int ctiles = cR / TC;
  const int tix = blk - p.block_start;
  const int ob = tix / ctiles, cb = tix - ob * ctiles;
  (void)oR;
  const int row = TC * taps, rs = row + 1;
  // (eight elements per thread and trip, every load requested before the first LDS store: the one-element form was 36 dependent
  //  round trips per thread -- 49 us per repack launch of 82 MB)
  constexpr int DUMMY = 32 * (32 * 9 + 1);
  const int nelem = 32 * row;
  for (int base = 0; base < nelem; base += 256 * 8) {
    float v[8];
    int di[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 256 * u + (int)threadIdx.x;
      const bool in = idx < nelem;
      const int ii = in ? idx : 0;
      const int ol = ii / row, rem = ii - ol * row;
      const int o = ob * 32 + ol, c = cb * TC + rem / taps;
      const bool ok = in && o < p.O && c < p.C;
      const float t = p.w[ok ? ((long)o * p.C + cb * TC) * taps + rem : 0];
      v[u] = ok ? t : 0.f;
      di[u] = in ? ol * rs + rem : DUMMY;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[di[u]] = v[u];
  }
  __syncthreads();
  const int per_t = 32 * TC;
  for (int idx = threadIdx.x; idx < p.T * per_t; idx += 256) {
    const int t = idx / per_t, r2 = idx - t * per_t;
    const int tr = p.ky[t] * p.KW + p.kx[t];
    int ol, cl;
    if (p.transpose) { cl = r2 % TC; ol = r2 / TC; } else { ol = r2 & 31; cl = r2 >> 5; }
    const int o = ob * 32 + ol, c = cb * TC + cl;
    const float v = tile[ol * rs + cl * taps + tr];
    if (p.transpose) p.dst[((long)t * p.KP + o) * p.MP + c] = v;
    else p.dst[((long)t * p.KP + c) * p.MP + o] = v;
  }
}

void launch_pack_jobs(const PackJob* d_jobs, int njobs, int total_blocks, hipStream_t s) {
  if (njobs == 0) return;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(total_blocks), dim3(256), 0, s, d_jobs, njobs);
  DBM_HIP(hipGetLastError());
}
