// Weight-gradient of the convolutions (backward of the L.Convolution2D call sites listed in
// igemm.hip) on the fp32 matrix cores:  gW[o][c][t] += scale * sum_{n,a,b} dy[n][o][a][b] * x[n][c][tap t].
//
// GEMM view: M = out channels, N = (in channel, tap), K = output positions.  The contiguous
// memory axis (positions) is the MFMA K axis, which per-lane global loads cannot feed
// efficiently (every lane would touch its own cache line), so operands are staged through LDS:
// for one "band" of positions a 32 x BP slab of dy and the zero-framed, tap-ready input planes;
// each wavefront owns one 32-channel input tile and keeps one 32x32 accumulator per tap, i.e.
// T independent MFMA chains fed by 1 + 2/T LDS dwords per MFMA.
//
// One layer alone cannot fill 256 CUs (the trunk's weight matrices have 2..12 32x32 tiles), and
// splitting K harder only multiplies the fp32 atomics that fold the partial sums.  So launches
// are BATCHED: a device-resident table of per-layer plans, one workgroup = (layer, input-channel
// group, output tile, K split); all weight gradients of a group of layers with the same kernel
// form go out in a single launch (they do not depend on each other).
//
// Three kernel forms (WgradBatch::build picks per layer):
//   wgrad_wave_dma_kernel   3x3 / stride 1 on small planes (the 9x9 trunk): whole-image bands, LDS-DMA staging,
//                           two wavefronts = two input tiles sharing one dy slab
//   wgrad_band_dma_kernel   3x3 / 4x4, any stride / folded resize, on large planes: row bands, dword LDS-DMA
//   wgrad_kernel            general form (1x1 layers, 4x4 layers on tiny planes): four / eight wavefronts,
//                           register-staged
// (later: wgrad_direct_kernel / wgrad_1x1_kernel for large planes, wgrad_s2tiny_kernel for the deep discriminator layers --
// 4x4 stride 2 AND, round 3, 3x3 stride 1 on planes of <= 4 x 4 -- each described where it is defined)
#include "dbm_internal.h"
#include <algorithm>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Phase timing of the LDS-DMA kernels for tools/wgrad_bench/ (compiled out of the library).
#ifdef DBM_WG_TIMING
__device__ unsigned long long g_dbg[4 * 8192];
#define WG_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define WG_TACC(acc, a, b) acc += (b) - (a)
#else
#define WG_T(var)
#define WG_TACC(acc, a, b)
#endif

bool g_wgrad_deterministic = true;  // the reference trains with cudnn_deterministic = True (srgan_train.py:69)

// Deterministic mode: a wavefront's accumulators go out in register order (256-byte stores) to its slot of the partial
// buffer; wgrad_fold_kernel sums the K slices in order.
// gradient targets of K slice bz: gW / gb themselves, or a pair buffer
__device__ __forceinline__ void pair_targets(const WgradPlan& p, int bz, float*& gW, float*& gb) {
  gW = p.d.gW; gb = p.d.gb;
  if (p.pairW) {
    const int k = (bz >> 1) - p.pair_direct;
    if (k >= 0) {
      gW = p.pairW + (long)k * p.pair_stride;
      gb = gb ? gW + (p.pair_stride - p.d.Cout) : nullptr;
    }
  }
}

template <int TPW>
__device__ __forceinline__ void store_partial(const WgradPlan& p, int bz, int by, int grp, int slot, int lane, float scale,
                                              const f32x16 (&acc)[TPW]) {
  float* dst = p.partial + ((((long)bz * p.coutTiles + by) * p.groups + grp) * p.fold_slots + slot) * (TPW * 1024) + lane;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64] = scale * acc[t][r];
}

// TPW = taps per wavefront.  3x3 / 1x1: every wavefront keeps all T accumulators (TPW = T, 4 wavefronts, 256
// registers -> two workgroups share a CU: one stages while the other feeds the MFMA pipe).  4x4: 16 accumulators are
// 256 registers on their own, so the taps are split over two wavefronts per input tile (TPW = 8, 8 wavefronts).
template <int T, int TPW>
__global__ __launch_bounds__(256 * (T / TPW), 2) void wgrad_kernel(const WgradPlan* __restrict__ plans,
                                                                   const int* __restrict__ starts, int nplans) {
  constexpr int TG = T / TPW;       // tap groups
  constexpr int NW = 4 * TG;        // wavefronts per workgroup
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // ---- which layer does this workgroup belong to? (binary search in the prefix table) ----
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= wg) lo = mid; else hi = mid - 1;
  }
  const WgradPlan& p = plans[lo];
  const WgradDesc& d = p.d;
  int local = wg - starts[lo];
  const int bx = local % p.groups;            // input-channel group
  local /= p.groups;
  const int by = local % p.coutTiles;         // output-channel tile
  const int bz = local / p.coutTiles;         // K split

  float* ldsY = lds;                              // 32 * YS
  int* pixoff = (int*)(ldsY + 32 * p.YS);         // BPp: patch offset of each position
  int* pixinfo = pixoff + p.BPp;                  // BPp: ib | al << 8 | b << 20
  int* xinfo = pixinfo + p.BPp;                   // IB*ImgS: ib | ry << 8 | rx << 20
  float* ldsX = (float*)(xinfo + p.IB * p.ImgS);  // G*32 * XS
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int ct = wave & 3;                    // input tile of this wavefront inside the group
  const int tg = wave >> 2;                   // tap group of this wavefront
  const int cout0 = by * 32;
  const int cin0 = bx * p.G * 32;
  const int cin_w = cin0 + ct * 32;
  const bool wave_active = ct < p.G && cin_w < d.Cin;
  const int Hl = d.Hin << d.ups, Wl = d.Win << d.ups;
  const int ROW = p.R * d.OW;

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  constexpr int KWc = (T == 1) ? 1 : (T == 9 ? 3 : 4);
  int toff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tt = tg * TPW + t;
    toff[t] = (tt / KWc) * p.Wst + (tt % KWc);
  }

  // ---- per-band-shape tables (no integer division in the staging loops) ----
  for (int e = tid; e < p.BPp; e += NT) {
    int off = 0, info = 0xff;  // ib = 255 marks the padding position
    if (e < p.BP) {
      const int ib = e / ROW;
      const int rem = e - ib * ROW;
      const int al = rem / d.OW, b = rem - al * d.OW;
      off = ib * p.ImgS + al * d.stride * p.Wst + b * d.stride;
      info = ib | (al << 8) | (b << 20);
    }
    pixoff[e] = off;
    pixinfo[e] = info;
  }
  for (int e = tid; e < p.IB * p.ImgS; e += NT) {
    const int ib = e / p.ImgS;
    const int rem = e - ib * p.ImgS;
    const int ry = rem / p.Wst, rx = rem - ry * p.Wst;
    xinfo[e] = ib | (ry << 8) | (rx << 20);
  }
  const int perch = p.IB * p.ImgS;
  const int nch = p.G * 32;
  if (p.fast) {
    for (int e = tid; e < nch * p.XS; e += NT) ldsX[e] = 0.f;
    for (int e = tid; e < 32 * p.YS; e += NT) ldsY[e] = 0.f;
  }

  for (int band = bz; band < p.nbands; band += p.S) {
    const int ig = band / p.nbr;
    const int a0 = (band - ig * p.nbr) * p.R;
    const int n0 = ig * p.IB;
    __syncthreads();  // tables written / previous band fully consumed
    // Staging is latency bound (few wavefronts per CU).  Each lane decodes ITS positions once per band and then
    // streams all channels / rows for them with U independent, unconditional loads in flight (out-of-image or
    // out-of-range elements read a safe address and are zeroed by a select afterwards).
    if (p.fast) {
      // Whole images of plain planes: per image the 32 dy planes and the nch input planes of this workgroup are ONE
      // contiguous run each.  16-byte loads, four in flight per lane, then scattered into the padded LDS planes
      // (the zero border was written once before the band loop and is never overwritten).
      const int oplane = d.OH * d.OW;
      const int nco = min(32, d.Cout - cout0);
      const int nci = min(nch, d.Cin - cin0);
      const int plane = d.Hin * d.Win;
      for (int ib = 0; ib < p.IB; ++ib) {
        const int n = n0 + ib;
        const bool okn = n < d.N;
        {  // dy: nco x oplane floats -> ldsY[i * YS + ib * oplane + pos]
          const int q4 = (nco * oplane) >> 2;
          const float4* src = (const float4*)(d.dy + (okn ? (long)n * d.dysn : 0L) + (long)cout0 * d.dysc);
          const unsigned oM = p.winM;  // (unused for dy) keep the register pressure flat
          (void)oM;
          for (int i = tid; i < q4; i += NT) {
            float4 v = src[i];
            if (!okn) v = make_float4(0.f, 0.f, 0.f, 0.f);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int f = 4 * i + u;
              const int row = f / oplane;  // oplane is not a compile-time constant: one division per element (few per band)
              ldsY[row * p.YS + ib * oplane + (f - row * oplane)] = vv[u];
            }
          }
        }
        {  // x: nci x plane floats -> ldsX[c * XS + ib * ImgS + (y + pad) * Wst + (x + pad)]
          const int q4 = (nci * plane) >> 2;
          const float4* src = (const float4*)(d.x + (okn ? (long)n * d.xsn : 0L) + (long)cin0 * d.xsc);
          float* dstb = ldsX + ib * p.ImgS + d.pad * p.Wst + d.pad;
          for (int i0 = tid; i0 < q4; i0 += 4 * NT) {
            float4 v[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const int i = i0 + w * NT;
              v[w] = src[i < q4 ? i : 0];
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const int i = i0 + w * NT;
              if (i < q4) {
                const float vv[4] = {v[w].x, v[w].y, v[w].z, v[w].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  const unsigned f = 4u * (unsigned)i + u;
                  const unsigned c = __umulhi(f, p.planeM);
                  const unsigned r = f - c * (unsigned)plane;
                  const unsigned y = __umulhi(r, p.winM);
                  const unsigned xx = r - y * (unsigned)d.Win;
                  if ((int)(y + d.pad) < p.Rin && (int)(xx + d.pad) < p.Wst)
                    dstb[c * p.XS + y * p.Wst + xx] = okn ? vv[u] : 0.f;
                }
              }
            }
          }
        }
      }
    } else {
      constexpr int U = 8;
      constexpr int RPW = 32 / NW;  // dy rows per wavefront
      // ---- dy slab: 32 x BPp ----
      for (int p0 = 0; p0 < p.BPp; p0 += 64) {
        const int pix = p0 + lane;
        const bool inb = pix < p.BPp;
        const int info = pixinfo[inb ? pix : 0];
        const int ib = info & 0xff, al = (info >> 8) & 0xfff, b = (int)((unsigned)info >> 20);
        const int n = n0 + ib, a = a0 + al;
        const bool ok = inb && ib != 0xff && n < d.N && a < d.OH;
        const float* src = d.dy + (ok ? (long)n * d.dysn + a * d.OW + b : 0L);
        float v[RPW];
  #pragma unroll
        for (int u = 0; u < RPW; ++u) {
          const int i = wave * RPW + u;
          v[u] = src[(ok && cout0 + i < d.Cout) ? (long)(cout0 + i) * d.dysc : 0L];
        }
        if (inb) {
  #pragma unroll
          for (int u = 0; u < RPW; ++u) {
            const int i = wave * RPW + u;
            ldsY[i * p.YS + pix] = (ok && cout0 + i < d.Cout) ? v[u] : 0.f;
          }
        }
      }
      // ---- input patch in logical (upsampled, zero padded) coordinates; each wavefront stages nch / NW channels ----
      const int cpw = nch / NW;  // nch is a multiple of 32, NW of 4 or 8
      for (int e0 = 0; e0 < perch; e0 += 64) {
        const int e = e0 + lane;
        const bool inb = e < perch;
        const int info = xinfo[inb ? e : 0];
        const int ib = info & 0xff, ry = (info >> 8) & 0xfff, rx = (int)((unsigned)info >> 20);
        const int iy = a0 * d.stride - d.pad + ry, ix = rx - d.pad;
        const int n = n0 + ib;
        const bool ok = inb && n < d.N && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
        const float* src = d.x + (ok ? (long)n * d.xsn + (iy >> d.ups) * d.Win + (ix >> d.ups) : 0L);
        for (int cb = 0; cb < cpw; cb += U) {
          float v[U];
  #pragma unroll
          for (int u = 0; u < U; ++u) {
            const int ci = cin0 + wave * cpw + cb + u;
            v[u] = src[(ok && cb + u < cpw && ci < d.Cin) ? (long)ci * d.xsc : 0L];
          }
          if (inb) {
  #pragma unroll
            for (int u = 0; u < U; ++u) {
              const int c = wave * cpw + cb + u;
              if (cb + u < cpw) ldsX[c * p.XS + e] = (ok && cin0 + c < d.Cin) ? v[u] : 0.f;
            }
          }
        }
      }
    }
    __syncthreads();
    if (d.gb && bx == 0 && tid < 256) {  // bias gradient: 8 threads per slab row, folded with wavefront shuffles
      const float* row = ldsY + (tid >> 3) * p.YS;
      float part = 0.f;
      for (int pix = tid & 7; pix < p.BP; pix += 8) part += row[pix];
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      bsum += part;  // every thread of the 8-group holds the row total; only (tid & 7) == 0 publishes it
    }
    if (wave_active) {
      // K loop, software pipelined by one full step: the A value and the TPW patch values of step k+1 (and the patch
      // offset of step k+2) are in flight while the MFMAs of step k issue.  hipcc would otherwise sink every LDS read
      // to just before its MFMA (one exposed LDS latency per read); the scheduling barriers pin the order.
      const float* arow = ldsY + j * p.YS + kh;
      const float* xrow = ldsX + (ct * 32 + j) * p.XS;
      float av = arow[0];
      float bv[TPW];
      {
        const int off = pixoff[kh];
#pragma unroll
        for (int t = 0; t < TPW; ++t) bv[t] = xrow[off + toff[t]];
      }
      int off_n = pixoff[((2 < p.BPp) ? 2 : 0) + kh];
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the loop is entered with nothing pending
      for (int kp = 0; kp < p.BPp; kp += 2) {
        const int kn = (kp + 2 < p.BPp) ? kp + 2 : kp;
        const int kn2 = (kp + 4 < p.BPp) ? kp + 4 : kn;
        const float av_n = arow[kn];
        float bv_n[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) bv_n[t] = xrow[off_n + toff[t]];
        off_n = pixoff[kn2 + kh];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        av = av_n;
#pragma unroll
        for (int t = 0; t < TPW; ++t) bv[t] = bv_n[t];
      }
    }
  }

  // ---- fold the partial sums into gW[o][c][t] (t fastest).  A lane owns (o, c) pairs with a stride of T floats
  // between lanes, so direct atomics would touch a different cache line per lane; instead the wavefronts of one
  // input tile transpose 8 output rows at a time through LDS and issue the atomics over consecutive addresses. ----
  if (p.partial) {
    if (wave_active) store_partial<TPW>(p, bz, by, bx, wave, lane, d.scale, acc);
    if (d.gb && bx == 0 && tid < 256 && (tid & 7) == 0) p.partial_b[((long)bz * p.coutTiles + by) * 32 + (tid >> 3)] = d.scale * bsum;
    return;
  }
  float *gWt, *gbt;
  pair_targets(p, bz, gWt, gbt);
  __syncthreads();  // staging buffers are dead: reuse the LDS
  {
    constexpr int ROWF = 32 * T;  // floats of one output row of a 32-channel input tile
    float* tw = lds + ct * (8 * ROWF);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (wave_active) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            tw[(rr + 4 * kh) * ROWF + j * T + tg * TPW + t] = d.scale * acc[t][4 * q + rr];
      }
      __syncthreads();
      if (wave_active) {
        for (int e = tg * 64 + lane; e < 8 * ROWF; e += 64 * TG) {
          const int rl = e / ROWF;
          const int rem = e - rl * ROWF;
          const int o = cout0 + 8 * q + rl;
          const int c = cin_w + rem / T;
          if (o < d.Cout && c < d.Cin) atomicAdd(gWt + ((long)o * d.Cin + cin_w) * T + rem, tw[e]);
        }
      }
      __syncthreads();
    }
  }
  if (d.gb && bx == 0 && tid < 256 && (tid & 7) == 0 && cout0 + (tid >> 3) < d.Cout)
    atomicAdd(gbt + cout0 + (tid >> 3), d.scale * bsum);
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-task form with LDS-DMA staging: 3x3 / stride 1 / pad 1 layers on small planes (the 9x9 trunk).  No staging
// registers and no staging arithmetic: dy (32 planes, contiguous in memory) lands with `global_load_lds_dwordx4`;
// x lands with `global_load_lds_dword`, two instructions per channel plane, into a zero-framed plane of
// (H+2) x (W+1) cells (one zero row above and below, one zero column shared by neighbouring rows) so that the K loop
// needs no border logic.  Every load of a band is in flight at once.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128, 2) void wgrad_wave_dma_kernel(const WgradPlan* __restrict__ plans, const int* __restrict__ starts,
                                                               int nplans) {
  constexpr int T = 9;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= wg) lo = mid; else hi = mid - 1;
  }
  const WgradPlan& p = plans[lo];
  const WgradDesc& d = p.d;
  int local = wg - starts[lo];
  const int wave = threadIdx.x >> 6;
  const int ct = 2 * (local % p.groups) + wave;  // two wavefronts = two input tiles share one dy slab
  local /= p.groups;
  const int by = local % p.coutTiles;
  const int bz = local / p.coutTiles;
  const int H = d.OH, W = d.OW, W1 = W + 1;
  const int plane = H * W;          // == Hin * Win
  const int slab = 32 * plane;      // floats of one image's 32 dy planes
  const int PS = p.ImgS;            // (H + 2) * (W + 1): framed x plane
  float* ldsY = lds;                            // IB * slab
  float* ldsX0 = ldsY + p.IB * slab + 4;        // 2 x IB * 32 * PS, behind a 4-float zero guard (tap (0,0) of cell (0,0))
  float* ldsX = ldsX0 + wave * p.IB * 32 * PS;  // this wavefront's framed planes
  int* pinfo = (int*)(ldsX0 + 2 * p.IB * 32 * PS);  // BPp + 8: position -> dy offset | x offset of tap (0,0) << 16
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int j = lane & 31, kh = lane >> 5;
  const int cout0 = by * 32, cin_w = ct * 32;
  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  int toff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) toff[t] = (t / 3) * W1 + (t % 3);
  for (int e = tid; e < p.BPp + 8; e += 128) {
    int info = 0;  // padding positions: dy offset of position 0, x offset 0 -- their dy is zero (see below)
    if (e < p.BP) {
      const int ib = e / plane;
      const int rem = e - ib * plane;
      const int y = rem / W, x = rem - y * W;
      info = (ib * slab + rem) | ((ib * 32 * PS + y * W1 + x) << 16);  // x offset is relative to ldsX - 1
    }
    pinfo[e] = info;
  }
  const int tot = p.IB * slab + 4 + 2 * p.IB * 32 * PS;
  for (int e = tid; e < tot; e += 128) lds[e] = 0.f;  // frames, guard, and the planes of ragged tiles
  // per-lane source cell of the two DMA instructions that fill one framed plane (cells lane, lane + 64)
  int soff[2];
  bool sval[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = lane + 64 * k;
    const int yy = r / W1, xx = r - yy * W1;
    sval[k] = r < PS && yy >= 1 && yy <= H && xx < W;
    soff[k] = sval[k] ? (yy - 1) * W + xx : 0;
  }
  const int nco = min(32, d.Cout - cout0), nci = max(0, min(32, d.Cin - cin_w));
  const bool wave_active = nci > 0;
  const int q4y = (nco * plane) >> 2;
  float bsum = 0.f;
#ifdef DBM_WG_TIMING
  unsigned long long tS = 0, tK = 0;
#endif
  WG_T(t00);
  for (int band = bz; band < p.nbands; band += p.S) {
    const int n0 = band * p.IB;
    WG_T(tA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // LDS stores / reads of the previous round (both wavefronts) are done before the DMA lands
    for (int ib = 0; ib < p.IB; ++ib) {
      const int n = n0 + ib;
      float* dsty = ldsY + ib * slab;
      float* dstx = ldsX + ib * 32 * PS;
      if (n < d.N) {
        const float4* srcy = (const float4*)(d.dy + (long)n * d.dysn + (long)cout0 * d.dysc);
        for (int i0 = 64 * wave; i0 < q4y; i0 += 128)  // the two wavefronts split the shared slab
          if (i0 + lane < q4y) __builtin_amdgcn_global_load_lds(srcy + i0 + lane, (lds_ptr)(dsty + 4 * i0), 16, 0, 0);
        const float* srcx = d.x + (long)n * d.xsn + (long)cin_w * d.xsc;
        for (int c = 0; c < nci; ++c) {
          if (sval[0]) __builtin_amdgcn_global_load_lds(srcx + c * plane + soff[0], (lds_ptr)(dstx + c * PS), 4, 0, 0);
          if (sval[1]) __builtin_amdgcn_global_load_lds(srcx + c * plane + soff[1], (lds_ptr)(dstx + c * PS + 64), 4, 0, 0);
        }
      } else {  // ragged last band: this image does not exist
        for (int e = tid; e < slab; e += 128) dsty[e] = 0.f;
        for (int e = lane; e < 32 * PS; e += 64) dstx[e] = 0.f;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    WG_T(tB);
    WG_TACC(tS, tA, tB);
    if (d.gb && ct == 0) {  // bias gradient: lane (j, kh) sums every other position of dy plane j
      // (straight from the slab, eight independent reads in flight: the position table's dependent read per element made
      // this loop a third of a band's time for the wavefront that owns it)
      float part[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] = 0.f;
      for (int ib = 0; ib < p.IB; ++ib) {
        const float* yp = ldsY + ib * slab + j * plane;
        const int first = (kh + ib * plane) & 1;  // parity of this image's first position inside the band
        int e = first;
        for (; e + 14 < plane; e += 16) {
#pragma unroll
          for (int u = 0; u < 8; ++u) part[u] += yp[e + 2 * u];
        }
        for (; e < plane; e += 2) part[0] += yp[e];
      }
      bsum += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    if (wave_active) {
      // K loop, software pipelined by one full step: all eleven LDS reads of step k+1 (and the table entry of step
      // k+2) are in flight while the nine MFMAs of step k issue.  hipcc would sink every read to just before its MFMA
      // (one exposed LDS latency per read); the scheduling barriers pin the order.
      const float* arow = ldsY + j * plane;
      const float* xrow = ldsX - 1 + j * PS;
      // the odd padding position of a band has table entry 0: it re-reads position 0, so its A value is forced to 0
      const bool odd_tail = (p.BP & 1) != 0;
      // Two operand sets in ping-pong (no `cur = next` copies, which hipcc places behind the MFMA block together with
      // the wait for the reads they copy), and the LDS reads of step k+1 are dealt out BETWEEN the MFMAs of step k
      // (sched_group_barrier: one MFMA, one DS read, ...): their issue slots disappear in the matrix pipe's shadow.
      const int BPp = __builtin_amdgcn_readfirstlane(p.BPp);  // (uniform; read through a vector load: keep the loop scalar)
      int info = pinfo[kh];
      float a0 = (odd_tail && kh == 1 && BPp == 2) ? 0.f : arow[info & 0xffff];
      float b0[T], a1 = 0.f, b1[T];
#pragma unroll
      for (int t = 0; t < T; ++t) { b0[t] = xrow[(info >> 16) + toff[t]]; b1[t] = 0.f; }
      int i0 = pinfo[2 + kh], i1 = pinfo[4 + kh];  // table entries of steps 1 and 2: each is read two steps before its use
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the loop is entered with nothing pending
      auto half = [&](int kp, float& ac, float (&bc)[T], float& an, float (&bn)[T], int& ie) {
        // consumes (ac, bc) = step kp; fetches (an, bn) = step kp + 2 (table entry ie) and the entry of step kp + 6
        __builtin_amdgcn_sched_barrier(0);
        // (the padding position's A value is zeroed HERE, where the wait for `ac` is due anyway: placed behind the load
        // of `an` it cost a full `lgkmcnt(0)` -- one exposed LDS latency -- in every other step)
        if (odd_tail && kh == 1 && kp + 2 >= BPp) ac = 0.f;
        const int inf = ie;
        ie = pinfo[kp + 6 + kh];
        an = arow[inf & 0xffff];  // (past the end: the guard entries)
#pragma unroll
        for (int t = 0; t < T; ++t) bn[t] = xrow[(inf >> 16) + toff[t]];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac, bc[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // the remaining reads
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int kp = 0; kp < BPp; kp += 4) {
        half(kp, a0, b0, a1, b1, i0);
        if (kp + 2 < BPp) half(kp + 2, a1, b1, a0, b0, i1);
      }
    }
#ifdef DBM_WG_TIMING
    asm volatile("s_nop 0" ::"v"(acc[8][15]));
#endif
    WG_T(tC);
    WG_TACC(tK, tB, tC);
  }
  WG_T(tE);
  if (p.partial) {
    if (wave_active) store_partial<T>(p, bz, by, ct >> 1, wave, lane, d.scale, acc);
    if (d.gb && ct == 0) {
      bsum += __shfl_xor(bsum, 32, 64);
      if (kh == 0) p.partial_b[((long)bz * p.coutTiles + by) * 32 + j] = d.scale * bsum;
    }
    return;
  }
  float *gWt, *gbt;
  pair_targets(p, bz, gWt, gbt);
  __syncthreads();  // the slabs are dead: each wavefront transposes through its own piece of the LDS
  if (wave_active) {
    constexpr int ROWF = 32 * T;
    float* tw = lds + wave * (8 * ROWF);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) tw[(rr + 4 * kh) * ROWF + j * T + t] = d.scale * acc[t][4 * q + rr];
      __builtin_amdgcn_wave_barrier();
      for (int e = lane; e < 8 * ROWF; e += 64) {
        const int rl = e / ROWF;
        const int rem = e - rl * ROWF;
        const int o = cout0 + 8 * q + rl;
        const int c = cin_w + rem / T;
        if (o < d.Cout && c < d.Cin) atomicAdd(gWt + ((long)o * d.Cin + cin_w) * T + rem, tw[e]);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (d.gb && ct == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (kh == 0 && cout0 + j < d.Cout) atomicAdd(gbt + cout0 + j, d.scale * bsum);
  }
#ifdef DBM_WG_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  if (tid == 0 && blockIdx.x < 8192) {
    WG_T(tZ);
    g_dbg[4 * blockIdx.x] = tS; g_dbg[4 * blockIdx.x + 1] = tK; g_dbg[4 * blockIdx.x + 2] = tZ - tE; g_dbg[4 * blockIdx.x + 3] = tZ - t00;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Row-band LDS-DMA form: 3x3 and 4x4 layers on planes too large for whole-image bands (the 18x18 / 36x36 layers of
// the generator tail and of the discriminator), any stride, with or without the folded nearest x2 resize.
// A band = R output rows of one image.  Both operands land with `global_load_lds_dword` (per-lane source address,
// lane-linear destination): dy rows into 32 slabs of odd stride, x into zero-framed rows of Wl + 1 cells per channel
// (out-of-image rows are fetched from a device zero word; the shared zero column is never written).  Two
// wavefronts per workgroup share the dy slab: for 3x3 they own two input tiles, for 4x4 the two halves of the
// sixteen taps of one input tile (sixteen 32x32 accumulators do not fit one wavefront).
// ---------------------------------------------------------------------------------------------------------------
template <int T, int TPW>
__global__ __launch_bounds__(128, 2) void wgrad_band_dma_kernel(const WgradPlan* __restrict__ plans, const int* __restrict__ starts,
                                                                int nplans) {
  constexpr int TG = T / TPW;  // 1: wavefronts = input tiles, 2: wavefronts = tap halves
  constexpr int KWc = (T == 9) ? 3 : 4;
  constexpr int NI = 8;        // DMA instructions per channel band / dy row band (<= 512 cells)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= wg) lo = mid; else hi = mid - 1;
  }
  const WgradPlan& p = plans[lo];
  const WgradDesc& d = p.d;
  int local = wg - starts[lo];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = (TG == 1) ? 2 * (local % p.groups) + wave : (local % p.groups);
  const int tg = (TG == 1) ? 0 : wave;
  local /= p.groups;
  const int by = local % p.coutTiles;
  const int bz = local / p.coutTiles;
  const int j = lane & 31, kh = lane >> 5;
  const int cout0 = by * 32, cin_w = ct * 32;
  const int st = d.stride, Hl = d.Hin << d.ups, Wl = d.Win << d.ups;
  const int Wf = Wl + 1, Rin = p.Rin, PSb = p.XS, YS = p.YS;
  const int xcells = Rin * Wf, ycells = p.R * d.OW;
  float* ldsY = lds;                                   // 32 * YS
  float* ldsX0 = ldsY + 32 * YS + 4;                   // behind a 4-float zero guard
  float* ldsX = ldsX0 + ((TG == 1) ? wave * 32 * PSb : 0);
  int* pinfo = (int*)(ldsX0 + ((TG == 1) ? 2 : 1) * 32 * PSb);  // BPp + 8
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  int toff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tt = tg * TPW + t;
    toff[t] = (tt / KWc) * Wf + (tt % KWc);
  }
  for (int e = tid; e < p.BPp + 8; e += 128) {
    int info = p.BP;  // padding positions: the (never written, zero) cell behind the last dy position of a slab
    if (e < p.BP) {
      const int al = e / d.OW, b = e - al * d.OW;
      info = e | ((al * st * Wf + b * st) << 16);
    }
    pinfo[e] = info;
  }
  const int tot = 32 * YS + 4 + ((TG == 1) ? 2 : 1) * 32 * PSb;
  for (int e = tid; e < tot; e += 128) lds[e] = 0.f;
  // per-lane cells of the DMA instructions that fill one channel band of x / one slab row band of dy
  int xry[NI], xrx[NI];
  bool xok[NI], yok[NI];
  int yal[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int cell = lane + 64 * k;
    const int ry = cell / Wf, rx = cell - ry * Wf;
    xok[k] = cell < xcells && rx < Wl;   // the last cell of a framed row is the shared zero column
    xry[k] = ry;
    xrx[k] = rx >> d.ups;
    yok[k] = cell < ycells;
    yal[k] = cell / d.OW;
  }
  const bool wave_active = cin_w < d.Cin;
  float bsum = 0.f;
#ifdef DBM_WG_TIMING
  unsigned long long tS = 0, tK = 0;
#endif
  WG_T(t00);
  for (int band = bz; band < p.nbands; band += p.S) {
    WG_T(tA);
    const int n = band / p.nbr;
    const int a0 = (band - n * p.nbr) * p.R;
    const int iy0 = a0 * st - d.pad;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // the previous band is consumed (and the LDS stores before the first band have landed)
    // ---- dy: slab rows are split between the two wavefronts ----
    for (int i = wave; i < 32; i += 2) {
      const bool rowok = cout0 + i < d.Cout;
      const float* src = d.dy + (long)n * d.dysn + (long)(cout0 + (rowok ? i : 0)) * d.dysc + (long)a0 * d.OW;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        if (64 * k < ycells) {
          if (yok[k]) {
            const float* g = (rowok && a0 + yal[k] < d.OH) ? src + lane + 64 * k : p.zeros;
            __builtin_amdgcn_global_load_lds(g, (lds_ptr)(ldsY + i * YS + 64 * k), 4, 0, 0);
          }
        }
      }
    }
    // ---- x: TG == 1: each wavefront fills its own 32 channels; TG == 2: the two wavefronts share one tile ----
    {
      const int c_lo = (TG == 1) ? 0 : 16 * wave, c_hi = (TG == 1) ? 32 : 16 * wave + 16;
      const float* srcn = d.x + (long)n * d.xsn;
      int soff[NI];
      bool rok[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int iy = iy0 + xry[k];
        rok[k] = (unsigned)iy < (unsigned)Hl;
        soff[k] = rok[k] ? (iy >> d.ups) * d.Win + xrx[k] : 0;
      }
      if (wave_active) {
        for (int c = c_lo; c < c_hi; ++c) {
          const bool cok = cin_w + c < d.Cin;
          const float* srcc = srcn + (long)(cin_w + (cok ? c : 0)) * d.xsc;
#pragma unroll
          for (int k = 0; k < NI; ++k) {
            if (64 * k < xcells) {
              if (xok[k]) {
                const float* g = (cok && rok[k]) ? srcc + soff[k] : p.zeros;
                __builtin_amdgcn_global_load_lds(g, (lds_ptr)(ldsX + c * PSb + 64 * k), 4, 0, 0);
              }
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    WG_T(tB);
    WG_TACC(tS, tA, tB);
    if (d.gb && ct == 0 && tg == 0) {  // bias gradient: lane (j, kh) sums every other position of slab row j
      float part = 0.f;
      for (int pix = kh; pix < p.BP; pix += 2) part += ldsY[j * YS + pix];
      bsum += part;
    }
    if (wave_active) {
      // K loop, software pipelined by one full step (see wgrad_wave_dma_kernel)
      const float* arow = ldsY + j * YS;
      const float* xrow = ldsX - d.pad + j * PSb;
      const int BPp = __builtin_amdgcn_readfirstlane(p.BPp);
      int info = pinfo[kh];
      float a0 = arow[info & 0xffff], a1 = 0.f;
      float b0[TPW], b1[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) { b0[t] = xrow[(info >> 16) + toff[t]]; b1[t] = 0.f; }
      int i0 = pinfo[2 + kh], i1 = pinfo[4 + kh];  // table entries of steps 1 and 2: read two steps before their use
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the loop is entered with nothing pending
      auto half = [&](int kp, float& ac, float (&bc)[TPW], float& an, float (&bn)[TPW], int& ie) {
        __builtin_amdgcn_sched_barrier(0);
        const int inf = ie;
        ie = pinfo[kp + 6 + kh];
        an = arow[inf & 0xffff];
#pragma unroll
        for (int t = 0; t < TPW; ++t) bn[t] = xrow[(inf >> 16) + toff[t]];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac, bc[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int kp = 0; kp < BPp; kp += 4) {
        half(kp, a0, b0, a1, b1, i0);
        if (kp + 2 < BPp) half(kp + 2, a1, b1, a0, b0, i1);
      }
    }
#ifdef DBM_WG_TIMING
    asm volatile("s_nop 0" ::"v"(acc[TPW - 1][15]));
#endif
    WG_T(tC);
    WG_TACC(tK, tB, tC);
  }
  WG_T(tE);
  if (p.partial) {
    if (wave_active) store_partial<TPW>(p, bz, by, (TG == 1) ? (ct >> 1) : ct, wave, lane, d.scale, acc);
    if (d.gb && ct == 0 && tg == 0) {
      bsum += __shfl_xor(bsum, 32, 64);
      if (kh == 0) p.partial_b[((long)bz * p.coutTiles + by) * 32 + j] = d.scale * bsum;
    }
    return;
  }
  float *gWt, *gbt;
  pair_targets(p, bz, gWt, gbt);
  // ---- fold into gW[o][c][t] (t fastest): 8 output rows at a time through LDS, atomics over consecutive addresses ----
  __syncthreads();
  {
    constexpr int ROWF = 32 * T;
    float* tw = lds + ((TG == 1) ? wave * (8 * ROWF) : 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (wave_active) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) tw[(rr + 4 * kh) * ROWF + j * T + tg * TPW + t] = d.scale * acc[t][4 * q + rr];
      }
      if (TG == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
      if (wave_active) {
        for (int e = ((TG == 1) ? lane : tid); e < 8 * ROWF; e += ((TG == 1) ? 64 : 128)) {
          const int rl = e / ROWF;
          const int rem = e - rl * ROWF;
          const int o = cout0 + 8 * q + rl;
          const int c = cin_w + rem / T;
          if (o < d.Cout && c < d.Cin) atomicAdd(gWt + ((long)o * d.Cin + cin_w) * T + rem, tw[e]);
        }
      }
      if (TG == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    }
  }
  if (d.gb && ct == 0 && tg == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (kh == 0 && cout0 + j < d.Cout) atomicAdd(gbt + cout0 + j, d.scale * bsum);
  }
#ifdef DBM_WG_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  if (tid == 0 && blockIdx.x < 8192) {
    WG_T(tZ);
    g_dbg[4 * blockIdx.x] = tS; g_dbg[4 * blockIdx.x + 1] = tK; g_dbg[4 * blockIdx.x + 2] = tZ - tE; g_dbg[4 * blockIdx.x + 3] = tZ - t00;
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------
// Direct form: 3x3 / stride 1 (plain or on a nearest-x2 input) and 4x4 / stride 2 layers on planes of >= 16 columns
// (generator tail, discriminator stem).  No LDS staging at all: the position axis is the MFMA K axis, and ANY
// assignment of positions to (instruction, k) is valid as long as both operands use it -- so lane (j, kh) takes four
// CONSECUTIVE positions of one output row segment (b0 = 8 q + 4 kh .. + 3) and reads them with one 16-byte load from
// ITS OWN row of dy (out channel j) and, per kernel row, one run of 6 / 4 / 10 floats from ITS OWN plane of x (in
// channel j): four instructions per tap consume them, k = 0 from the lanes kh = 0, k = 1 from the lanes kh = 1.
// Per segment of 8 positions a wavefront issues 36 (32) MFMAs against 1 + 6 (5) loads; the loads of the next segment are
// in flight meanwhile.  Borders are selects on the loaded values (masks depend on indices only), out-of-range addresses
// are clamped (every tensor carries a 128-byte guard).  A workgroup = four wavefronts = four consecutive K sub-slices of
// one (layer, out tile, in tile, tap half); they are summed through LDS and leave as ONE contribution per K slice.
// ---------------------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) f32x4u { float v[4]; };
struct __attribute__((packed, aligned(4))) f32x2u { float v[2]; };

template <int MODE>  // 0: 3x3 s1 p1;  1: 3x3 s1 p1 on a nearest-x2 input;  2: 4x4 s2 p1 (one tap half per workgroup)
__global__ __launch_bounds__(256, 2) void wgrad_direct_kernel(const WgradPlan* __restrict__ plans, const int* __restrict__ starts,
                                                              int nplans) {
  constexpr int T = MODE == 2 ? 16 : 9;
  constexpr int TPW = MODE == 2 ? 8 : 9;
  constexpr int TG = T / TPW;
  constexpr int KWc = MODE == 2 ? 4 : 3;
  constexpr int NR = MODE == 2 ? 2 : 3;                       // kernel rows per wavefront
  constexpr int NB = MODE == 0 ? 6 : (MODE == 1 ? 4 : 10);     // floats per kernel row and lane
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= wg) lo = mid; else hi = mid - 1;
  }
  const WgradPlan& p = plans[lo];
  const WgradDesc& d = p.d;
  int local = wg - starts[lo];
  const int bx = local % p.groups; local /= p.groups;
  const int by = local % p.coutTiles; local /= p.coutTiles;
  const int tg = local % TG;
  const int bz = local / TG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int cout0 = by * 32, cin_w = bx * 32;
  const int SPR = p.Wst;                 // segments of 8 positions per output row
  const int nseg = p.nbands;             // N * OH * SPR
  const int per = (nseg + p.S - 1) / p.S;
  const int g0 = bz * per, g1 = min(nseg, g0 + per);
  const int perw = (max(g1 - g0, 0) + 3) >> 2;
  const int s0 = g0 + wave * perw, s1 = min(g1, s0 + perw);
  const int Hl = d.Hin << d.ups, Wl = d.Win << d.ups;
  const int oc = min(cout0 + j, d.Cout - 1), ic = min(cin_w + j, d.Cin - 1);
  const float* dyl = d.dy + (long)oc * d.dysc;
  const float* xl = d.x + (long)ic * d.xsc;

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const bool want_bias = d.gb && bx == 0 && tg == 0;

  struct Seg {
    float a[4];
    float b[NR][NB];
    unsigned am;        // valid positions (4 bits)
    unsigned rm;        // valid kernel rows (NR bits)
    unsigned cm;        // valid columns (NB bits)
  };
  auto load = [&](int g, Seg& sg) {
    const int row = g / SPR, q = g - row * SPR;
    const int n = row / d.OH, a = row - n * d.OH;
    const int b0 = 8 * q + 4 * kh;
    const f32x4u av = *reinterpret_cast<const f32x4u*>(dyl + (long)n * d.dysn + a * d.OW + b0);
#pragma unroll
    for (int m = 0; m < 4; ++m) sg.a[m] = av.v[m];
    sg.am = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) sg.am |= (b0 + m < d.OW) ? (1u << m) : 0u;
    const float* xn = xl + (long)n * d.xsn;
    sg.rm = 0;
    sg.cm = 0;
    int c0;   // first physical column of the run
    if (MODE == 0) c0 = b0 - 1;
    else if (MODE == 1) c0 = (b0 >> 1) - 1;
    else c0 = 2 * b0 - 1;
#pragma unroll
    for (int e = 0; e < (MODE == 1 ? 6 : NB); ++e) {
      // logical column of element e (MODE 1: six logical columns share four physical ones)
      const int ix = (MODE == 2 ? 2 * b0 : b0) - 1 + e;
      if ((unsigned)ix < (unsigned)Wl) sg.cm |= 1u << e;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int il = (MODE == 2) ? 2 * a + 2 * tg + r - 1 : a + r - 1;   // logical input row
      const bool ok = (unsigned)il < (unsigned)Hl;
      if (ok) sg.rm |= 1u << r;
      const int iy = (ok ? il : 0) >> d.ups;
      const float* src = xn + iy * d.Win + c0;
      const f32x4u v0 = *reinterpret_cast<const f32x4u*>(src);
#pragma unroll
      for (int e = 0; e < 4; ++e) sg.b[r][e] = v0.v[e];
      if (MODE == 0) {
        const f32x2u v1 = *reinterpret_cast<const f32x2u*>(src + 4);
        sg.b[r][4] = v1.v[0]; sg.b[r][5] = v1.v[1];
      } else if (MODE == 2) {
        const f32x4u v1 = *reinterpret_cast<const f32x4u*>(src + 4);
        const f32x2u v2 = *reinterpret_cast<const f32x2u*>(src + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) sg.b[r][4 + e] = v1.v[e];
        sg.b[r][8] = v2.v[0]; sg.b[r][9] = v2.v[1];
      }
    }
  };
  auto compute = [&](const Seg& sg) {
    float a[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = ((sg.am >> m) & 1u) ? sg.a[m] : 0.f;
    if (want_bias) bsum += (a[0] + a[1]) + (a[2] + a[3]);
    // logical-column view of the runs, borders zeroed
    constexpr int NL = MODE == 1 ? 6 : NB;
    float bl[NR][NL];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int e = 0; e < NL; ++e) {
        const int pe = MODE == 1 ? (((e - 1) >> 1) + 1) : e;   // physical element of logical column e
        const bool ok = ((sg.rm >> r) & 1u) && ((sg.cm >> e) & 1u);
        bl[r][e] = ok ? sg.b[r][pe] : 0.f;
      }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int r = t / KWc, kx = t % KWc;
        const int e = (MODE == 2 ? 2 * m : m) + kx;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bl[r][e], acc[t], 0, 0, 0);
      }
  };

  if (s0 < s1) {
    Seg sa, sb;  // ping-pong (a `cur = nxt` copy waits for the loads it copies: see igemm.hip)
    load(s0, sa);
    for (int g = s0; g < s1; g += 2) {
      load(g + 1 < s1 ? g + 1 : g, sb);
      __builtin_amdgcn_sched_barrier(0);
      compute(sa);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < s1) {
        load(g + 2 < s1 ? g + 2 : g + 1, sa);
        __builtin_amdgcn_sched_barrier(0);
        compute(sb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- four K sub-slices -> one contribution: 8 output rows at a time through LDS, atomics over consecutive addresses ----
  float *gWt, *gbt;
  pair_targets(p, bz, gWt, gbt);
  constexpr int ROWF = 32 * T;
  float* tw = lds + wave * (8 * ROWF);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) tw[(rr + 4 * kh) * ROWF + j * T + tg * TPW + t] = acc[t][4 * q + rr];
    __syncthreads();
    for (int e = tid; e < 8 * ROWF; e += 256) {
      const int rl = e / ROWF;
      const int rem = e - rl * ROWF;
      const int c = rem / T, t = rem - c * T;
      if (TG > 1 && (t / TPW) != tg) continue;
      const int o = cout0 + 8 * q + rl;
      if (o < d.Cout && cin_w + c < d.Cin) {
        const float v = ((lds[e] + lds[8 * ROWF + e]) + lds[2 * 8 * ROWF + e]) + lds[3 * 8 * ROWF + e];
        atomicAdd(gWt + ((long)o * d.Cin + cin_w) * T + rem, d.scale * v);
      }
    }
    __syncthreads();
  }
  if (want_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (kh == 0) lds[wave * 32 + j] = bsum;
    __syncthreads();
    if (tid < 32 && cout0 + tid < d.Cout)
      atomicAdd(gbt + cout0 + tid, d.scale * (((lds[tid] + lds[32 + tid]) + lds[64 + tid]) + lds[96 + tid]));
  }
}


// ---------------------------------------------------------------------------------------------------------------
// 1x1 layers on large contiguous planes (the deformable convolution's 576 -> 64 GEMM over the sampled columns,
// srgan_train.py:506-523): gW[o][c] = sum over positions of dy[o][pos] * x[c][pos], K = N * plane positions.  Both
// operands are K-contiguous in memory, so they are staged through LDS in full 128-byte rows (16-byte loads and stores),
// double buffered: the loads of band k + 1 are in flight while band k feeds the matrix pipe.  A workgroup owns
// 64 (out) x 128 (in) of the gradient: wavefront w the 32 input channels 32 w .. 32 w + 31 against BOTH output tiles
// (two accumulators, three ds_read_b128 per eight MFMAs); a lane's 16-byte read covers four consecutive positions of its
// row -- any position-to-(instruction, k) assignment is valid as long as both operands use it.  Row stride 36 floats:
// 16-byte aligned and conflict-free for the 16-lane groups of ds_read_b128.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_1x1_kernel(const WgradPlan* __restrict__ plans, const int* __restrict__ starts,
                                                           int nplans) {
  constexpr int BP = 32, BS = 36;             // positions per band, LDS row stride
  constexpr int ROWS = 64 + 128;              // dy rows, then x rows
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x ROWS x BS
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= wg) lo = mid; else hi = mid - 1;
  }
  const WgradPlan& p = plans[lo];
  const WgradDesc& d = p.d;
  int local = wg - starts[lo];
  const int bx = local % p.groups; local /= p.groups;     // 128 input channels
  const int by = local % p.coutTiles;                     // 64 output channels
  const int bz = local / p.coutTiles;                     // K slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int cout0 = by * 64, cin0 = bx * 128;
  const int plane = d.OH * d.OW;
  const int bpi = p.nbr;                                  // bands per image
  const int per = (p.nbands + p.S - 1) / p.S;
  const int k0 = bz * per, k1 = min(p.nbands, k0 + per);
  // staging: thread `tid` moves rows r0 + 32 i (i < 6), positions 4 * (tid & 7) .. + 3
  const int sp = 4 * (tid & 7), r0 = tid >> 3;
  const float* src[6];
  bool rok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int r = r0 + 32 * i;
    if (r < 64) {
      rok[i] = cout0 + r < d.Cout;
      src[i] = d.dy + (long)(rok[i] ? cout0 + r : 0) * d.dysc;
    } else {
      rok[i] = cin0 + r - 64 < d.Cin;
      src[i] = d.x + (long)(rok[i] ? cin0 + r - 64 : 0) * d.xsc;
    }
  }
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 stage[6];
  auto issue = [&](int k) {
    const int n = k / bpi, pos = (k - n * bpi) * BP + sp;
    const bool pok = pos < plane;   // (planes are multiples of 4 positions: a 16-byte run is all in or all out)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const long img = (long)n * (i < 2 ? d.dysn : d.xsn);
      stage[i] = *reinterpret_cast<const f4*>(src[i] + img + (pok ? pos : 0));
      if (!(pok && rok[i])) stage[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto commit = [&](int buf) {
    float* base = lds + buf * (ROWS * BS);
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f4*>(base + (r0 + 32 * i) * BS + sp) = stage[i];
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float bs0 = 0.f, bs1 = 0.f;
  const bool want_bias = d.gb && bx == 0 && wave == 0;
  if (k0 < k1) {
    issue(k0);
    commit(0);
    __syncthreads();
    for (int k = k0; k < k1; ++k) {
      const int buf = (k - k0) & 1;
      if (k + 1 < k1) issue(k + 1);
      __builtin_amdgcn_sched_barrier(0);
      const float* A = lds + buf * (ROWS * BS) + j * BS + 4 * kh;
      const float* B = lds + buf * (ROWS * BS) + (64 + 32 * wave + j) * BS + 4 * kh;
#pragma unroll
      for (int u = 0; u < BP / 8; ++u) {
        const f4 a0 = *reinterpret_cast<const f4*>(A + 8 * u);
        const f4 a1 = *reinterpret_cast<const f4*>(A + 32 * BS + 8 * u);
        const f4 b = *reinterpret_cast<const f4*>(B + 8 * u);
        if (want_bias) { bs0 += (a0.x + a0.y) + (a0.z + a0.w); bs1 += (a1.x + a1.y) + (a1.z + a1.w); }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k + 1 < k1) commit(buf ^ 1);
      __syncthreads();  // band k + 1 is in place; every wavefront is done with band k's buffer
    }
  }
  // ---- fold: gW[o][c] rows are contiguous along c, the accumulators' lane axis: coalesced atomics, no transpose ----
  float *gWt, *gbt;
  pair_targets(p, bz, gWt, gbt);
  const int c = cin0 + 32 * wave + j;
  if (c < d.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = cout0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (o < d.Cout) atomicAdd(gWt + (long)o * d.Cin + c, d.scale * acc0[r]);
      if (o + 32 < d.Cout) atomicAdd(gWt + (long)(o + 32) * d.Cin + c, d.scale * acc1[r]);
    }
  }
  if (want_bias) {
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (kh == 0) {
      if (cout0 + j < d.Cout) atomicAdd(gbt + cout0 + j, d.scale * bs0);
      if (cout0 + 32 + j < d.Cout) atomicAdd(gbt + cout0 + 32 + j, d.scale * bs1);
    }
  }
}


// Sums the K-slice partials of a weight-gradient launch in slice order and adds them to gW (OIHW) / gb: one workgroup per
// 256 elements of one wavefront slot's tile.  No fp32 atomics on the K split (the one atomic per element below only
// serialises the real- and the fake-batch graph of the discriminator: two contributions onto a cleared gradient, and
// a + b == b + a), hence bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const WgradPlan* __restrict__ plans, const int* __restrict__ fstarts,
                                                         int nplans) {
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (fstarts[mid] <= wg) lo = mid; else hi = mid - 1;
  }
  const WgradPlan& p = plans[lo];
  const WgradDesc& d = p.d;
  const int T = d.KH * d.KW;
  const int parts = p.fold_tpw * 4;  // 256-element pieces of a slot's fold_tpw * 1024 floats
  int local = wg - fstarts[lo];
  const int part = local % parts; local /= parts;
  const int slot = local % p.fold_slots; local /= p.fold_slots;
  const int grp = local % p.groups;
  const int by = local / p.groups;
  const int ctl = slot % p.fold_cts, tg = slot / p.fold_cts;
  const int ct = grp * p.fold_ctmul + ctl;
  const int cout0 = by * 32, cin_w = ct * 32;
  if (d.gb && grp == 0 && slot == 0 && part == 0 && threadIdx.x < 32 && cout0 + (int)threadIdx.x < d.Cout) {
    float v = 0.f;
    for (int z = 0; z < p.S; ++z) v += p.partial_b[((long)z * p.coutTiles + by) * 32 + threadIdx.x];
    atomicAdd(d.gb + cout0 + threadIdx.x, v);
  }
  if (ctl >= p.fold_ctmul || cin_w >= d.Cin) return;
  const long tsz = (long)p.fold_tpw * 1024;
  const long tile = (((long)by * p.groups + grp) * p.fold_slots + slot) * tsz;
  const long sstride = (long)p.coutTiles * p.groups * p.fold_slots * tsz;
  const int e = part * 256 + threadIdx.x;
  float v = 0.f;
#pragma unroll 4
  for (int z = 0; z < p.S; ++z) v += p.partial[z * sstride + tile + e];
  const int lane = e & 63, r = (e >> 6) & 15, t = tg * p.fold_tpw + (e >> 10);
  const int o = cout0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  const int c = cin_w + (lane & 31);
  if (o < d.Cout && c < d.Cin) atomicAdd(d.gW + ((long)o * d.Cin + c) * T + t, v);
}


// ---------------------------------------------------------------------------------------------------------------
// 4x4 / stride 2 / pad 1 layers on tiny output planes (OW <= 4: the discriminator's conv_layer5 / 7 / 9, 9x9 -> 4x4,
// 4x4 -> 2x2, 2x2 -> 1x1) and, round 3, 3x3 / stride 1 / pad 1 layers on such planes (conv_layer6 / 8: they used to go
// through the trunk's LDS-DMA form, whose whole-image bands pay for all nine taps of every position; here output rows whose
// input row is padding are never enumerated: 267 -> 164 us for the tail's weight gradients, with the LDS table sized by the
// launch instead of a static 64 KB).  GEMM view per tap: M = out channels, N = in channels, K = (image, output position) -- a few
// hundred at batch 64, against 16 taps x (Cout / 32) x (Cin / 32) output tiles: output-bound.  No LDS staging and no K
// split: a workgroup owns one 16 (out) x 16 (in) tile (v_mfma_f32_16x16x4_f32: four times the workgroups of 32 x 32
// tiles, which these layers need -- conv_layer5 has 64 of those) for ALL sixteen taps, wavefront w the kernel row ky = w
// with one accumulator per kx.  Lane (j, k4) takes K entry 4 s + k4 of the row's valid (image, a, b) list; its A value is one
// dword of dy (out channel j), its four B values -- the taps kx = 0..3 at input columns 2 b - 1 .. 2 b + 2 -- are ONE
// 16-byte load from its own plane of x (in channel j; out-of-image columns are zeroed after the load, every tensor
// carries a 128-byte guard).  Output rows whose input row 2 a + ky - 1 falls outside the image are not enumerated at
// all.  Both graphs of the discriminator step (real and fake batch) are processed by the SAME workgroup, one after
// the other, into the same accumulators: the gradient leaves with a plain read-modify-write of elements nobody else
// touches -- no atomics, no partial buffers, bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------------
struct TinyPlan {
  const float* x[2];    // input activations of the (up to two) graphs; x[1] == null: one graph
  const float* dy[2];
  float* gW;
  long xsn, dysn;
  int N, Cin, Cout, Hin, Win, OH, OW;
  float scale;
  int wg_start, ctiles;  // workgroups of this layer: (Cout / 16) * ctiles, ctiles = Cin / 16
  int K;                 // 4: 4x4 stride 2 pad 1;  3: 3x3 stride 1 pad 1 (round 3: conv_layer6 / 8 of the discriminator)
};

struct __attribute__((packed, aligned(4))) f32x4u4 { float v[4]; };
constexpr int TINY_MAXK = 2048;  // K entries (images x valid output positions) a kernel row may have: 64 KB of LDS tables

typedef float f32x4 __attribute__((ext_vector_type(4)));

// KS x KS taps, stride ST, pad 1; wavefront ky < KS owns kernel row ky (a 3x3 layer leaves the workgroup's fourth wavefront idle)
template <int KS, int ST>
__device__ __forceinline__ void tiny_body(const TinyPlan& p, int local, int2* tb, int ky, int lane) {
  const int ot = local / p.ctiles, ctile = local - ot * p.ctiles;   // 16 x 16 tiles of (out, in) channels
  const int j = lane & 15, k4 = lane >> 4;
  const int OH = p.OH, OW = p.OW, Hin = p.Hin, Win = p.Win;
  // output rows a whose input row ST a + ky - 1 lies inside the image
  const int a_lo = ky == 0 ? 1 : 0;
  int a_hi = Hin - ky < 0 ? -1 : (Hin - ky) / ST;  // ST a + ky - 1 <= Hin - 1
  if (a_hi > OH - 1) a_hi = OH - 1;
  const int OHv = a_hi - a_lo + 1;
  f32x4 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (OHv > 0) {
    const int per_img = OHv * OW;
    const int E = p.N * per_img;               // K entries of this kernel row
    const int steps = (E + 3) >> 2;            // four per MFMA (v_mfma_f32_16x16x4_f32)
    const int oplane = OH * OW, iplane = Hin * Win;
    // The (image, a, b) decode of every K entry -- divisions, 64-bit products -- is done ONCE per wavefront into an LDS
    // table (the first version decoded per fetch: 2000 VALU instructions per 64 MFMAs, five times the MFMA time);
    // both graphs use it.  Entries past the end read offset 0 with an all-zero mask.
    for (int e = lane; e < 4 * steps; e += 64) {
      int2 v = make_int2(0, 0);
      if (e < E) {
        const int n = e / per_img;
        const int rem = e - n * per_img;
        const int ai = rem / OW, b = rem - ai * OW;
        const int a = a_lo + ai;
        const int row = ST * a + ky - 1, col0 = ST * b - 1;
        unsigned cm = 0;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) cm |= ((unsigned)(col0 + kx) < (unsigned)Win) ? (1u << kx) : 0u;
        v.x = (int)((long)n * p.xsn + row * Win + col0);
        v.y = (int)((unsigned)((long)n * p.dysn + a * OW + b) | (cm << 27) | (1u << 31));
      }
      tb[e] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wavefront wrote the table it reads (no other does)
    __builtin_amdgcn_wave_barrier();
    auto fetch = [&](const float* xl, const float* dyl, int s, float& av, f32x4u4& bv, unsigned& cm) {
      const int2 t = tb[4 * s + k4];
      cm = (unsigned)t.y >> 27;  // bit 4: valid entry
      av = dyl[(unsigned)t.y & 0x7ffffffu];
      bv = *reinterpret_cast<const f32x4u4*>(xl + t.x);  // (3x3: the fourth word is not used -- guard bytes at the very end of a tensor)
    };
    // Chunks of eight K steps in ping-pong: the sixteen loads of chunk q + 1 (scattered: every lane reads its own plane)
    // are in flight underneath the 8 KS MFMAs of chunk q.
    constexpr int CH = 8;
    struct Chunk { float a[CH]; f32x4u4 b[CH]; unsigned m[CH]; };
    auto fetch_chunk = [&](const float* xl, const float* dyl, int s0, Chunk& c) {
#pragma unroll
      for (int u = 0; u < CH; ++u) fetch(xl, dyl, s0 + u < steps ? s0 + u : steps - 1, c.a[u], c.b[u], c.m[u]);
    };
    auto mfma_chunk = [&](const Chunk& c, int s0) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        // (steps past the end re-read the last one, entries past the end read element 0: their A value is zeroed)
        const float av = (s0 + u < steps && (c.m[u] & 16u)) ? c.a[u] : 0.f;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
          acc[kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ((c.m[u] >> kx) & 1u) ? c.b[u].v[kx] : 0.f, acc[kx], 0, 0, 0);
      }
    };
    for (int g = 0; g < 2; ++g) {
      if (!p.x[g]) break;
      const float* xl = p.x[g] + (long)(ctile * 16 + j) * iplane;   // this lane's input plane / output-gradient plane
      const float* dyl = p.dy[g] + (long)(ot * 16 + j) * oplane;
      Chunk c0, c1;
      fetch_chunk(xl, dyl, 0, c0);
      for (int s = 0; s < steps; s += 2 * CH) {
        fetch_chunk(xl, dyl, s + CH, c1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(c0, s);
        __builtin_amdgcn_sched_barrier(0);
        if (s + CH < steps) {
          fetch_chunk(xl, dyl, s + 2 * CH, c0);
          __builtin_amdgcn_sched_barrier(0);
          mfma_chunk(c1, s + CH);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // gW[o][c][ky][0..KS) += scale * acc: KS words per (o, c), this workgroup is their only writer.
  // D layout of v_mfma_f32_16x16x4_f32: register r of lane (j, k4) is row 4 k4 + r (out channel), column j (in channel)
  float* gbase = p.gW + ((long)(ot * 16) * p.Cin + ctile * 16 + j) * (KS * KS) + ky * KS;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * k4 + r;
    float* dst = gbase + (long)o * p.Cin * (KS * KS);  // (4-byte alignment is all the arena guarantees)
    if constexpr (KS == 4) {
      f32x4u4 v = *reinterpret_cast<f32x4u4*>(dst);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) v.v[kx] += p.scale * acc[kx][r];
      *reinterpret_cast<f32x4u4*>(dst) = v;
    } else {
      float v[KS];
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) v[kx] = dst[kx];
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) dst[kx] = v[kx] + p.scale * acc[kx][r];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_s2tiny_kernel(const TinyPlan* __restrict__ plans, int nplans, int rowlen) {
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (plans[mid].wg_start <= wg) lo = mid; else hi = mid - 1;
  }
  const TinyPlan& p = plans[__builtin_amdgcn_readfirstlane(lo)];  // (uniform: the plan's fields become scalar loads)
  const int local = wg - p.wg_start;
  const int lane = threadIdx.x & 63, ky = threadIdx.x >> 6;
  // per kernel row: K entry -> {x offset, dy offset | column mask << 27 | valid << 31}; `rowlen` entries per row = the longest
  // row of the launch's layers (round 3: it was a static 4 x 2048 table, 64 KB -- two workgroups per CU; the step's layers need
  // 1028 entries, 32 KB: five)
  extern __shared__ int2 tbl[];
  if (p.K == 4) tiny_body<4, 2>(p, local, tbl + ky * rowlen, ky, lane);
  else if (ky < 3) tiny_body<3, 1>(p, local, tbl + ky * rowlen, ky, lane);
}

static bool s2tiny_eligible(const WgradDesc& d) {
  static const int on = DBM_TUNE_GETENV("WGRAD_TINY") ? atoi(DBM_TUNE_GETENV("WGRAD_TINY")) : 1;   // bit 0: 4x4 stride 2, bit 1: 3x3 stride 1
  const bool k4 = (on & 1) && d.KH == 4 && d.KW == 4 && d.stride == 2;
  static const int on3 = DBM_TUNE_GETENV("WGRAD_TINY3") ? atoi(DBM_TUNE_GETENV("WGRAD_TINY3")) : 1;
  const bool k3 = on3 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.Hin == d.OH && d.Win == d.OW;
  return (k4 || k3) && d.pad == 1 && d.ups == 0 && d.OW <= 4 && d.OH <= 4 && d.gb == nullptr &&
         d.Cin % 32 == 0 && d.Cout % 32 == 0 && d.xsc == d.Hin * d.Win && d.dysc == d.OH * d.OW && d.Win >= 2 &&
         (long)d.N * d.OH * d.OW + 4 <= TINY_MAXK && (long)d.N * d.xsn < (1L << 31) && (long)d.N * d.dysn < (1L << 27);
}

static inline int odd_up(int v) { return v | 1; }

// The other deterministic folding (pair buffers, see WgradPlan::pairW): gW += buffers in order, the buffers are cleared for
// the next launch.  Elementwise and coalesced: the buffers have gW's layout.
__global__ __launch_bounds__(256) void wgrad_pair_fold_kernel(const WgradPlan* __restrict__ plans, const int* __restrict__ fstarts,
                                                              int nplans) {
  int lo = 0, hi = nplans - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (fstarts[mid] <= wg) lo = mid; else hi = mid - 1;
  }
  const WgradPlan& p = plans[lo];
  if (!p.pairW || p.pair_n == 0) return;
  const long wsize = p.pair_stride - p.d.Cout;
  const long i = (long)(wg - fstarts[lo]) * 256 + threadIdx.x;
  if (i >= p.pair_stride) return;
  // (every buffer is read before the first one is cleared: `v += *q; *q = 0` per buffer put each load behind the previous store --
  //  vmcnt is in order over both, so every iteration waited for a write round trip)
  float v = 0.f;
  const int n = p.pair_n;
  for (int k0 = 0; k0 < n; k0 += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = p.pairW[(long)(k0 + u < n ? k0 + u : k0) * p.pair_stride + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += k0 + u < n ? t[u] : 0.f;   // (buffer order: the sum is the one the sequential loop formed)
  }
  for (int k = 0; k < n; ++k) p.pairW[(long)k * p.pair_stride + i] = 0.f;
  if (i < wsize) atomicAdd(p.d.gW + i, v);
  else if (p.d.gb) atomicAdd(p.d.gb + (i - wsize), v);
}

static size_t band_plan(const WgradDesc& d, WgradPlan& p, int S_fixed) {
  // row-band LDS-DMA form (wgrad_band_dma_kernel): the largest band of R output rows whose slabs fit half a CU's LDS
  const int T = d.KH * d.KW;
  if (!(T == 9 || T == 16) || d.KH != d.KW) return 0;
  const int tiles = (d.Cin + 31) / 32;
  p.groups = T == 9 ? (tiles + 1) / 2 : tiles;
  p.G = T == 9 ? 2 : 1;
  p.coutTiles = (d.Cout + 31) / 32;
  const int nx = T == 9 ? 2 : 1;
  const int Wl = d.Win << d.ups, Wf = Wl + 1;
  auto floats = [&](int R) {
    const int Rin = (R - 1) * d.stride + d.KH;
    const int bpp = (R * d.OW + 1) & ~1;
    return 32L * odd_up(bpp) + 4 + (long)nx * 32 * odd_up(Rin * Wf) + bpp + 8;
  };
  auto fits = [&](int R) {
    const int Rin = (R - 1) * d.stride + d.KH;
    return Rin * Wf <= 512 && R * d.OW <= 512 && floats(R) <= 19400 && (R * d.stride + d.KH) * Wf < 32768;
  };
  if (!fits(1) || d.OH * d.OW < 64) return 0;  // tiny planes: several images per band (workgroup / trunk forms)
  int R = 1;
  while (R < d.OH && fits(R + 1)) ++R;
  // equal bands: the smallest R with the same number of bands per image
  const int nbr = (d.OH + R - 1) / R;
  R = (d.OH + nbr - 1) / nbr;
  p.IB = 1; p.R = R; p.nbr = nbr;
  p.BP = R * d.OW;
  p.BPp = (p.BP + 1) & ~1;
  p.YS = odd_up(p.BPp);
  p.Rin = (R - 1) * d.stride + d.KH;
  p.Wst = Wf;
  p.ImgS = p.Rin * Wf;
  p.XS = odd_up(p.ImgS);
  p.nbands = d.N * nbr;
  int S = S_fixed > 0 ? S_fixed : 1;
  if (S > p.nbands) S = p.nbands;
  p.S = S;
  p.wg_count = p.groups * p.coutTiles * S;
  p.fast = 0; p.planeM = p.winM = p.oplaneM = 0;
  const size_t stage = sizeof(float) * (size_t)floats(R);
  return std::max(stage, sizeof(float) * 2 * 8 * 32 * (size_t)9);
}

// direct form (wgrad_direct_kernel): mode 0 / 1 / 2 or -1 if the layer is not eligible
static int direct_mode(const WgradDesc& d) {
  const int T = d.KH * d.KW;
  const int Hl = d.Hin << d.ups, Wl = d.Win << d.ups;
  if (d.KH != d.KW || d.pad != 1 || d.OW < 16 || d.xsc != d.Hin * d.Win || d.dysc != d.OH * d.OW) return -1;
  if (T == 9 && d.stride == 1 && d.OH == Hl && d.OW == Wl) return d.ups ? 1 : 0;
  if (T == 16 && d.stride == 2 && d.ups == 0 && 2 * d.OH == Hl && 2 * d.OW == Wl) return 2;
  return -1;
}

// segs_per_wg: K segments (8 positions each) one workgroup (four wavefronts) works through
static size_t direct_plan(const WgradDesc& d, WgradPlan& p, long segs_per_wg) {
  const int mode = direct_mode(d);
  DBM_CHECK(mode >= 0, "wgrad: layer not eligible for the direct form");
  const int T = d.KH * d.KW;
  p.d = d;
  p.wave_task = 4;
  p.zeros = nullptr;
  p.groups = (d.Cin + 31) / 32;
  p.G = 1;
  p.coutTiles = (d.Cout + 31) / 32;
  p.Wst = (d.OW + 7) / 8;
  const long nseg = (long)d.N * d.OH * p.Wst;
  DBM_CHECK(nseg < (1L << 30), "wgrad: too many positions");
  p.nbands = (int)nseg;
  long S = (nseg + segs_per_wg - 1) / segs_per_wg;
  if (S < 1) S = 1;
  p.S = (int)S;
  p.wg_count = p.groups * p.coutTiles * (mode == 2 ? 2 : 1) * p.S;
  p.IB = 1; p.R = 1; p.nbr = d.OH; p.BP = p.BPp = 8; p.YS = p.XS = 0; p.Rin = 0; p.ImgS = 0;
  p.fast = 0; p.planeM = p.winM = p.oplaneM = 0;
  return sizeof(float) * 4 * 8 * 32 * (size_t)T;  // four transpose areas of 8 output rows
}

// 1x1 layers on large contiguous planes (wgrad_1x1_kernel)
static bool gemm1x1_eligible(const WgradDesc& d) {
  const int plane = d.OH * d.OW;
  return d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 && d.ups == 0 && d.Hin == d.OH && d.Win == d.OW && plane >= 256 &&
         plane % 4 == 0 && d.xsc == plane && d.dysc == plane && d.xsn % 4 == 0 && d.dysn % 4 == 0 &&
         ((uintptr_t)d.x % 16) == 0 && ((uintptr_t)d.dy % 16) == 0;
}

static size_t gemm1x1_plan(const WgradDesc& d, WgradPlan& p, long bands_per_wg) {
  p.d = d;
  p.wave_task = 5;
  p.zeros = nullptr;
  p.groups = (d.Cin + 127) / 128;
  p.G = 4;
  p.coutTiles = (d.Cout + 63) / 64;
  const int plane = d.OH * d.OW;
  p.nbr = (plane + 31) / 32;
  p.nbands = d.N * p.nbr;
  long S = (p.nbands + bands_per_wg - 1) / bands_per_wg;
  if (S < 1) S = 1;
  p.S = (int)S;
  p.wg_count = p.groups * p.coutTiles * p.S;
  p.IB = 1; p.R = 1; p.BP = p.BPp = 32; p.YS = p.XS = 36; p.Rin = 0; p.ImgS = 0; p.Wst = 0;
  p.fast = 0; p.planeM = p.winM = p.oplaneM = 0;
  return sizeof(float) * 2 * (64 + 128) * 36;
}

size_t wgrad_plan(const WgradDesc& d, WgradPlan& p, int level, int wave_task, int S_fixed) {
  const int T = d.KH * d.KW;
  DBM_CHECK(T == 1 || T == 9 || T == 16, "wgrad: supported kernels are 1x1, 3x3, 4x4");
  DBM_CHECK(d.OW < 4096 && d.OH < 4096, "wgrad: image too large");
  p.d = d;
  p.wave_task = wave_task;
  p.zeros = nullptr;
  if (wave_task == 3) return band_plan(d, p, S_fixed);
  const bool dma = wave_task == 2;
  if (dma && !(T == 9 && d.stride == 1 && d.pad == 1 && d.OH == d.Hin && d.OW == d.Win && (d.OH + 2) * (d.OW + 1) <= 128)) return 0;
  const int tiles = (d.Cin + 31) / 32;
  p.groups = dma ? (tiles + 1) / 2 : (tiles + 3) / 4;
  p.G = dma ? 2 : (tiles + p.groups - 1) / p.groups;
  p.coutTiles = (d.Cout + 31) / 32;
  // band selection: whole images if small, else row bands of one image
  const int Wst = (d.OW - 1) * d.stride + d.KW;
  // floats; workgroup form: <= 79 KB lets two workgroups share a CU's 160 KB; dma: four two-wavefront groups per CU
  const long budget = dma ? 10000 : (T <= 9) ? 19800 : 25000;
  auto cost = [&](int IB, int R) {
    const int Rin = (R - 1) * d.stride + d.KH;
    const long bpp = (IB * R * d.OW + 1) & ~1;
    if (dma) return 32L * IB * d.OH * d.OW + 4 + 64L * IB * (d.OH + 2) * (d.OW + 1) + bpp + 8;
    return (long)p.G * 32 * odd_up(IB * Rin * Wst) + 32L * odd_up((int)bpp) + 2 * bpp + (long)IB * Rin * Wst;
  };
  int IB = 1, R = d.OH;
  if (cost(1, d.OH) <= budget) {
    while (IB * 2 <= d.N && IB * 2 <= 128 && IB * 2 * d.OH * d.OW <= (324 >> std::max(level, 0)) && cost(IB * 2, d.OH) <= budget) IB *= 2;
  } else {
    if (dma) return 0;
    while (R > 1 && cost(1, R) > budget) --R;
  }
  DBM_CHECK(cost(IB, R) <= 38000, "wgrad: band does not fit in LDS");
  p.IB = IB; p.R = R;
  p.nbr = (d.OH + R - 1) / R;
  p.BP = IB * R * d.OW;
  p.BPp = (p.BP + 1) & ~1;
  p.YS = odd_up(p.BPp);
  p.Rin = (R - 1) * d.stride + d.KH;
  p.Wst = Wst;
  p.ImgS = p.Rin * p.Wst;
  p.XS = odd_up(IB * p.ImgS);
  DBM_CHECK(p.Rin < 4096 && p.Wst < 4096, "wgrad: patch too large");
  const int imgGroups = (d.N + IB - 1) / IB;
  p.nbands = imgGroups * p.nbr;
  // K split: enough positions per workgroup that the closing atomics stay a small fraction of the MFMA work
  // (~8 images of the 9x9 trunk per workgroup; `level` > 0 halves that, and the band, per step: used by batches
  // that would otherwise leave most of the chip idle)
  const long positions = (long)d.N * d.OH * d.OW;
  const long per_wg = level >= 0 ? std::max(1L, 648L >> level) : (648L << -level);
  int S = (int)((positions + per_wg - 1) / per_wg);
  if (S_fixed > 0) S = S_fixed;
  if (S > p.nbands) S = p.nbands;
  if (S < 1) S = 1;
  p.S = S;
  p.wg_count = p.groups * p.coutTiles * S;
  auto magic = [](unsigned dv) { return (unsigned)((0x100000000ULL + dv - 1) / dv); };
  const int plane = d.Hin * d.Win, oplane = d.OH * d.OW;
  p.planeM = magic((unsigned)plane);
  p.winM = magic((unsigned)d.Win);
  p.oplaneM = magic((unsigned)oplane);
  // contiguous 16-byte staging: whole-image bands of plain planes, every per-image run 16-byte aligned and a
  // multiple of four floats (all channel counts of a run: full tiles / groups and the ragged last one)
  const int chx = p.G * 32;
  const int tail_ci = d.Cin - (p.groups - 1) * chx, tail_co = d.Cout - (p.coutTiles - 1) * 32;
  auto mult4 = [](long v) { return (v % 4) == 0; };
  // (d.Win > 1: the multiply-high constants ceil(2^32 / dv) do not exist for dv = 1 -- a plane one pixel wide, e.g. the 1x1
  // view of the input block's k6 s2 branch on 3 x 3 tiles, got every sample of a run decoded to channel 0; found by the
  // randomised-geometry tests of round 3)
  p.fast = d.Win > 1 && p.nbr == 1 && d.ups == 0 && d.xsc == plane && d.dysc == oplane && mult4(d.xsn) && mult4(d.dysn) &&
           ((uintptr_t)d.x % 16) == 0 && ((uintptr_t)d.dy % 16) == 0 && mult4(32L * oplane) && mult4((long)chx * plane) &&
           mult4((long)std::min(32, tail_co) * oplane) && mult4((long)std::min(chx, tail_ci) * plane) &&
           (long)chx * plane < (1L << 20) && 32L * oplane < (1L << 20) && plane < 4096 && oplane < 4096;
  if (dma) {
    if (!p.fast || (long)IB * 32 * oplane >= 65536 || (long)IB * 32 * (d.OH + 2) * (d.OW + 1) >= 32768) return 0;
    p.ImgS = (d.OH + 2) * (d.OW + 1);
    const size_t stage = sizeof(float) * ((size_t)32 * IB * oplane + 4 + (size_t)64 * IB * p.ImgS + (size_t)p.BPp + 8);
    return std::max(stage, sizeof(float) * 2 * 8 * 32 * (size_t)T);
  }
  const size_t stage = sizeof(float) * ((size_t)32 * p.YS + 2 * (size_t)p.BPp + (size_t)IB * p.ImgS + (size_t)p.G * 32 * p.XS);
  const size_t epilogue = sizeof(float) * 4 * 8 * 32 * (size_t)T;  // per-wavefront transpose areas
  return std::max(stage, epilogue);
}

template <int T, int TPW>
static void launch_T(const WgradPlan* plans, const int* starts, int nplans, int total_wg, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<T, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<T, TPW>), dim3(total_wg), dim3(256 * (T / TPW)), lds, s, plans, starts, nplans);
  DBM_HIP(hipGetLastError());
}

void WgradBatch::reset() {
  descs.clear();
  built = false;
  if (d_tiny) (void)hipFree(d_tiny);
  d_tiny = nullptr;
  n_tiny = tiny_wgs = 0;
  tiny_flops = 0.0;
  tiny_owner.clear();
  for (int g = 0; g < NCAT; ++g) {
    if (d_plans[g]) (void)hipFree(d_plans[g]);
    if (d_starts[g]) (void)hipFree(d_starts[g]);
    if (d_partial[g]) (void)hipFree(d_partial[g]);
    d_partial[g] = nullptr;
    fold_wgs[g] = 0;
    d_plans[g] = nullptr;
    d_starts[g] = nullptr;
    nplans[g] = total_wg[g] = 0;
    lds[g] = 0;
    flops[g] = 0.0;
  }
}

static int direct_form_enabled() {  // DBM_WGRAD_DIRECT: 1 (default) = wgrad_direct_kernel for the large-plane 3x3 / 4x4 layers
  static int v = -1;
  if (v < 0) {
    const char* e = DBM_TUNE_GETENV("WGRAD_DIRECT");
    v = e ? atoi(e) : 1;
  }
  return v;
}

static int dma_forms_enabled() {  // DBM_WGRAD_WAVE: 0 = workgroup form only, 1 = + trunk LDS-DMA tasks, 2 (default) = + row-band LDS-DMA
  static int v = -1;
  if (v < 0) {
    const char* e = DBM_TUNE_GETENV("WGRAD_WAVE");
    v = e ? atoi(e) : 2;
  }
  return v;
}

static const float* device_zeros() {
  static float* z = nullptr;
  if (!z) {
    DBM_HIP(hipMalloc((void**)&z, 256));
    DBM_HIP(hipMemset(z, 0, 256));
    DBM_HIP(hipDeviceSynchronize());
  }
  return z;
}

void WgradBatch::build() {
  // categories: 0 = 1x1, 1 = 3x3 workgroup form, 2 = 4x4 workgroup form, 3 = 3x3 trunk LDS-DMA tasks (whole-image bands
  // of small planes), 4 = 3x3 row-band LDS-DMA, 5 = 4x4 row-band LDS-DMA, 6 / 7 / 8 = direct form (3x3, 3x3 on a
  // nearest-x2 input, 4x4 stride 2)
  // 9 = 1x1 on large contiguous planes (LDS-staged GEMM, wgrad_1x1_kernel)
  static const int TT[NCAT] = {1, 9, 16, 9, 9, 16, 9, 9, 16, 1};
  static const int MODE[NCAT] = {0, 0, 0, 2, 3, 3, 4, 4, 4, 5};
  const bool direct = direct_form_enabled() != 0;
  std::vector<int> cat(descs.size());
  const int forms = dma_forms_enabled();
  // ---- 4x4 stride-2 layers on tiny planes: their own kernel (wgrad_s2tiny_kernel), outside the category tables.  Two
  // descriptors with the same gradient (the real- and the fake-batch graph of the discriminator) share a plan. ----
  {
    std::vector<TinyPlan> tp;
    std::vector<int> owner(descs.size(), -1);
    int wgs = 0;
    double fl = 0.0, tiny_bytes_acc = 0.0;
    for (size_t i = 0; i < descs.size(); ++i) {
      if (owner[i] >= 0) continue;
      const WgradDesc& d = descs[i];
      // the descriptors that add to this gradient: all of them must qualify, at most two, same geometry
      std::vector<size_t> grp;
      bool ok = true;
      for (size_t k = 0; k < descs.size(); ++k) {
        if (descs[k].gW != d.gW) continue;
        const WgradDesc& f = descs[k];
        grp.push_back(k);
        ok = ok && s2tiny_eligible(f) && f.KH == d.KH && f.N == d.N && f.Cin == d.Cin && f.Cout == d.Cout && f.Hin == d.Hin && f.Win == d.Win &&
             f.OH == d.OH && f.OW == d.OW && f.xsn == d.xsn && f.dysn == d.dysn && f.scale == d.scale;
      }
      if (!ok || grp.size() > 2 || grp[0] != i) continue;
      TinyPlan q;
      memset(&q, 0, sizeof(q));
      for (size_t u = 0; u < grp.size(); ++u) { q.x[u] = descs[grp[u]].x; q.dy[u] = descs[grp[u]].dy; owner[grp[u]] = (int)tp.size(); }
      q.gW = d.gW; q.xsn = d.xsn; q.dysn = d.dysn;
      q.N = d.N; q.Cin = d.Cin; q.Cout = d.Cout; q.Hin = d.Hin; q.Win = d.Win; q.OH = d.OH; q.OW = d.OW; q.scale = d.scale;
      q.ctiles = d.Cin / 16; q.wg_start = wgs; q.K = d.KH;
      wgs += (d.Cout / 16) * q.ctiles;
      tp.push_back(q);
      fl += 2.0 * (double)grp.size() * d.N * d.OH * d.OW * d.Cout * d.Cin * (d.KH * d.KW);
      tiny_bytes_acc += 4.0 * ((double)grp.size() * d.N * ((double)d.Cin * d.Hin * d.Win + (double)d.Cout * d.OH * d.OW) + (double)d.Cout * d.Cin * (d.KH * d.KW));
    }
    n_tiny = (int)tp.size(); tiny_wgs = wgs; tiny_flops = fl; tiny_bytes = tiny_bytes_acc;
    tiny_rowlen = 4;
    for (const TinyPlan& q : tp) tiny_rowlen = std::max(tiny_rowlen, ((q.N * q.OH * q.OW + 3) / 4) * 4);
    tiny_owner = owner;
    if (d_tiny) { (void)hipFree(d_tiny); d_tiny = nullptr; }
    if (n_tiny) {
      DBM_HIP(hipMalloc((void**)&d_tiny, tp.size() * sizeof(TinyPlan)));
      DBM_HIP(hipMemcpy(d_tiny, tp.data(), tp.size() * sizeof(TinyPlan), hipMemcpyHostToDevice));
    }
  }
  for (size_t i = 0; i < descs.size(); ++i) {
    if (tiny_owner[i] >= 0) { cat[i] = -1; continue; }
    const int T = descs[i].KH * descs[i].KW;
    WgradPlan p;
    const int dm = direct ? direct_mode(descs[i]) : -1;
    if (T == 1) cat[i] = (direct && gemm1x1_eligible(descs[i])) ? 9 : 0;
    else if (T == 9) cat[i] = (forms >= 1 && wgrad_plan(descs[i], p, 0, 2) != 0) ? 3 : dm >= 0 ? 6 + dm : (forms >= 2 && wgrad_plan(descs[i], p, 0, 3) != 0) ? 4 : 1;
    else cat[i] = dm >= 0 ? 6 + dm : (forms >= 2 && wgrad_plan(descs[i], p, 0, 3) != 0) ? 5 : 2;
  }
  for (int g = 0; g < NCAT; ++g) {
    std::vector<WgradPlan> plans;
    std::vector<int> starts;
    int total = 0;
    size_t maxlds = 0;
    double fl = 0.0, by = 0.0;
    // LDS-DMA forms: the K split is chosen so that the launch is ONE round of equally long two-wavefront groups
    // (four per CU for the trunk form; two or four per CU for row bands, by their LDS footprint)
    int S_fixed = 0;
    long segs_per_wg = 0;
    if (g == 9) {
      long work = 0;
      for (size_t i = 0; i < descs.size(); ++i) {
        if (cat[i] != g) continue;
        const WgradDesc& d = descs[i];
        work += (long)((d.Cin + 127) / 128) * ((d.Cout + 63) / 64) * d.N * ((d.OH * d.OW + 31) / 32);
      }
      static const int slots_env = DBM_TUNE_GETENV("WGRAD_1X1_WGS") ? atoi(DBM_TUNE_GETENV("WGRAD_1X1_WGS")) : 512;
      segs_per_wg = std::max(8L, (work + slots_env - 1) / slots_env);   // bands of 32 positions per workgroup
    } else if (g >= 6) {
      // direct form: equally long workgroups, about two per CU over the whole launch (a workgroup's K range should stay
      // long enough that its closing atomics are a small fraction: >= 64 segments = 512 positions)
      long work = 0;
      for (size_t i = 0; i < descs.size(); ++i) {
        if (cat[i] != g) continue;
        const WgradDesc& d = descs[i];
        work += (long)((d.Cin + 31) / 32) * ((d.Cout + 31) / 32) * (g == 8 ? 2 : 1) * d.N * d.OH * ((d.OW + 7) / 8);
      }
      static const int slots_env = DBM_TUNE_GETENV("WGRAD_DIRECT_WGS") ? atoi(DBM_TUNE_GETENV("WGRAD_DIRECT_WGS")) : 512;
      segs_per_wg = std::max(64L, (work + slots_env - 1) / slots_env);
    } else if (g >= 3) {
      long units = 0;
      size_t need = 0;
      for (size_t i = 0; i < descs.size(); ++i) {
        if (cat[i] != g) continue;
        WgradPlan p;
        need = std::max(need, wgrad_plan(descs[i], p, 0, MODE[g], 1));
        units += (long)p.groups * p.coutTiles;
      }
      static const int slots_env = DBM_TUNE_GETENV("WGRAD_SLOTS") ? atoi(DBM_TUNE_GETENV("WGRAD_SLOTS")) : 0;
      // (few-layer launches -- the discriminator's conv_layer4: 16 units -- take a coarser K split: 256 workgroups of four images
      // instead of 1024 of one, a quarter of the partial tiles to write and fold: 116 -> 90 us standalone, round 3)
      static const int slots_small = DBM_TUNE_GETENV("WGRAD_SLOTS_SMALL") ? atoi(DBM_TUNE_GETENV("WGRAD_SLOTS_SMALL")) : 256;
      int slots = slots_env ? slots_env : (need > 40 * 1024 ? 512 : 1024);
      // (<= 32 units: in data-parallel runs the trunk's launches are cut into four groups of 126 units each -- those keep the fine split)
      if (g == 3 && slots_small && units > 0 && units <= 32) slots = std::min(slots, slots_small);  // (the row-band forms lose: 164 -> 204 us)
      if (units > 0) S_fixed = (int)std::max(1L, slots / units);
    }
    // workgroup forms: a launch should offer about two workgroups per CU; small batches split their position axis finer
    for (int level = 0; level < 4; ++level) {
      plans.clear(); starts.clear();
      total = 0; maxlds = 0; fl = 0.0;
      for (size_t i = 0; i < descs.size(); ++i) {
        if (cat[i] != g) continue;
        const WgradDesc& d = descs[i];
        WgradPlan p;
        maxlds = std::max(maxlds, g == 9 ? gemm1x1_plan(d, p, segs_per_wg) : g >= 6 ? direct_plan(d, p, segs_per_wg)
                                                                                    : wgrad_plan(d, p, g >= 3 ? 0 : level, MODE[g], S_fixed));
        p.zeros = device_zeros();
        p.partial = nullptr; p.partial_b = nullptr; p.fold_start = 0;
        p.pairW = nullptr; p.pair_n = 0; p.pair_direct = 0; p.pair_stride = 0;
        p.fold_slots = p.fold_cts = p.fold_ctmul = p.fold_tpw = 0;
        starts.push_back(total);
        total += p.wg_count;
        plans.push_back(p);
        fl += 2.0 * (double)d.N * d.OH * d.OW * d.Cout * d.Cin * TT[g];
        by += 4.0 * ((double)d.N * ((double)d.Cin * d.Hin * d.Win + (double)d.Cout * d.OH * d.OW) + (double)d.Cout * d.Cin * TT[g]);
      }
      // (deterministic mode: no finer K split than necessary -- every extra slice is a pair buffer to write and fold --
      // but a launch of a few dozen workgroups, e.g. the 4x4 layers of the deep discriminator, is split as well)
      // (256 since round 5 -- the input block's GEMM-shaped launch, 144 workgroups at 128, ends the iteration behind the trunk's launch:
      //  62.6 -> 46 us; 7.61-7.64 against 7.62-7.68 ms per step, 448: 7.64-7.67, 64: 7.68-7.71)
      static const int det_min = DBM_TUNE_GETENV("WGRAD_DET_MINWG") ? atoi(DBM_TUNE_GETENV("WGRAD_DET_MINWG")) : 256;
      if (total >= 448 || plans.empty() || g >= 3 || (g_wgrad_deterministic && total >= det_min)) break;
    }
    std::vector<int> fstarts;
    fold_wgs[g] = 0;
    if (g_wgrad_deterministic && !plans.empty()) {
      static const int SLOTS[NCAT] = {4, 4, 8, 2, 2, 2, 0, 0, 0, 0}, CTS[NCAT] = {4, 4, 4, 2, 2, 1, 0, 0, 0, 0},
                       TPWS[NCAT] = {1, 9, 8, 9, 9, 8, 0, 0, 0, 0};
      static const int pairs_env = DBM_TUNE_GETENV("WGRAD_PAIRS") ? atoi(DBM_TUNE_GETENV("WGRAD_PAIRS")) : 1;
      pair_mode[g] = pairs_env != 0 || g >= 6;  // (the direct form folds through pair buffers only)
      size_t floats = 0, bfloats = 0;
      int fw = 0;
      if (pair_mode[g]) {
        // pair buffers: slices 2k, 2k + 1 -> buffer k (pair 0 straight to the gradient when it is known to be zero)
        for (auto& pl : plans) {
          const int npair = (pl.S + 1) / 2, direct = cleared_target ? 1 : 0;
          const long stride = (((long)pl.d.Cout * pl.d.Cin * pl.d.KH * pl.d.KW + pl.d.Cout) + 3) & ~3L;
          if (pl.S > 1 && npair - direct > 0) floats += (size_t)(npair - direct) * stride;
        }
        if (d_partial[g]) (void)hipFree(d_partial[g]);
        DBM_HIP(hipMalloc((void**)&d_partial[g], (floats + 4) * sizeof(float)));
        DBM_HIP(hipMemset(d_partial[g], 0, (floats + 4) * sizeof(float)));
        float* base = d_partial[g];
        for (auto& pl : plans) {
          pl.fold_start = fw;
          fstarts.push_back(fw);
          pl.pairW = nullptr; pl.pair_n = 0; pl.pair_direct = 0; pl.pair_stride = 0;
          if (pl.S <= 1) continue;
          const int npair = (pl.S + 1) / 2, direct = cleared_target ? 1 : 0;
          if (npair - direct <= 0) continue;  // two slices onto a cleared gradient: plain atomics commute
          // (the stride is the exact tensor size + bias; the arena offset is rounded up to four floats)
          pl.pair_stride = (long)pl.d.Cout * pl.d.Cin * pl.d.KH * pl.d.KW + pl.d.Cout;
          pl.pairW = base; pl.pair_n = npair - direct; pl.pair_direct = direct;
          base += (size_t)pl.pair_n * ((pl.pair_stride + 3) & ~3L);
          fw += (int)((pl.pair_stride + 255) / 256);
        }
      } else {
      // a layer without a K split needs no partials: its single contribution per launch goes out with the atomic
      // epilogue (at most two launches' worth meet on a cleared gradient: still order-independent)
      for (auto& pl : plans) {
        pl.pairW = nullptr; pl.pair_n = 0; pl.pair_direct = 0; pl.pair_stride = 0;
        if (pl.S <= (cleared_target ? 2 : 1)) continue;
        floats += (size_t)pl.S * pl.coutTiles * pl.groups * SLOTS[g] * TPWS[g] * 1024;
        bfloats += (size_t)pl.S * pl.coutTiles * 32;
      }
      if (d_partial[g]) (void)hipFree(d_partial[g]);
      DBM_HIP(hipMalloc((void**)&d_partial[g], (floats + bfloats + 1) * sizeof(float)));
      DBM_HIP(hipMemset(d_partial[g], 0, (floats + bfloats + 1) * sizeof(float)));  // (inactive slots are never written)
      float* base = d_partial[g];
      float* bbase = base + floats;
      for (auto& pl : plans) {
        pl.fold_start = fw;
        fstarts.push_back(fw);
        if (pl.S <= (cleared_target ? 2 : 1)) continue;  // (an empty range in the fold table)
        pl.partial = base; pl.partial_b = bbase;
        pl.fold_slots = SLOTS[g]; pl.fold_cts = CTS[g]; pl.fold_tpw = TPWS[g];
        pl.fold_ctmul = (g <= 2) ? pl.G : (g == 5 ? 1 : 2);
        base += (size_t)pl.S * pl.coutTiles * pl.groups * SLOTS[g] * TPWS[g] * 1024;
        bbase += (size_t)pl.S * pl.coutTiles * 32;
        fw += pl.coutTiles * pl.groups * SLOTS[g] * TPWS[g] * 4;
      }
      }
      fold_wgs[g] = fw;
      for (int v : fstarts) starts.push_back(v);  // the fold table rides behind the launch table
    }
    starts.insert(starts.begin() + (long)plans.size(), total);  // (prefix table of the launch: plans.size() + 1 entries)
    nplans[g] = (int)plans.size();
    total_wg[g] = total;
    lds[g] = maxlds;
    flops[g] = fl;
    abytes[g] = by;
    if (plans.empty()) continue;
    DBM_HIP(hipMalloc((void**)&d_plans[g], plans.size() * sizeof(WgradPlan)));
    DBM_HIP(hipMalloc((void**)&d_starts[g], starts.size() * sizeof(int)));
    DBM_HIP(hipMemcpy(d_plans[g], plans.data(), plans.size() * sizeof(WgradPlan), hipMemcpyHostToDevice));
    DBM_HIP(hipMemcpy(d_starts[g], starts.data(), starts.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  DBM_HIP(hipDeviceSynchronize());
  built = true;
  built_deterministic = g_wgrad_deterministic;
  built_cleared = cleared_target;
}

template <typename K>
static void launch_dma(K kernel, const WgradPlan* plans, const int* starts, int nplans, int total_wg, size_t lds, hipStream_t s,
                       int threads = 128) {
  static bool attr_set = false;
  if (!attr_set) {  // (one flag for the three kernels: they share this instantiation's signature)
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_wave_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_band_dma_kernel<9, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_band_dma_kernel<16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_direct_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_direct_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_direct_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)wgrad_1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kernel, dim3(total_wg), dim3(threads), lds, s, plans, starts, nplans);
  DBM_HIP(hipGetLastError());
}

void WgradBatch::launch(hipStream_t s) {
  // libdbm_measure.so only (tools/phases.py): the data-gradient chains alone (gradients are then wrong)
  static const bool skip_all = DBM_MEASURE_ENV("NO_WGRAD") != 0;
  if (skip_all) return;
  if (built && (built_deterministic != g_wgrad_deterministic || built_cleared != cleared_target)) {  // mode switched: re-plan (keeps the descriptors)
    std::vector<WgradDesc> keep = descs;
    reset();
    descs = keep;
  }
  if (!built) build();
  if (n_tiny) {
    if (g_profiler.enabled) g_profiler.begin(s, 1, tiny_flops, tiny_bytes, "wgrad_s2tiny", tiny_wgs);
    hipLaunchKernelGGL(wgrad_s2tiny_kernel, dim3(tiny_wgs), dim3(256), 4 * (size_t)tiny_rowlen * sizeof(int2), s, (const TinyPlan*)d_tiny, n_tiny,
                       tiny_rowlen);
    DBM_HIP(hipGetLastError());
    if (g_profiler.enabled) g_profiler.end(s);
  }
  for (int g = 0; g < NCAT; ++g) {
    if (nplans[g] == 0) continue;
    if (g_profiler.enabled) {
      static const char* const cat_name[NCAT] = {"wgrad<1,1>", "wgrad<9,9>", "wgrad<16,8>", "wave_dma", "band_dma<9,9>", "band_dma<16,8>",
                                                 "direct<0>", "direct<1>", "direct<2>", "wgrad_1x1"};
      char tag[40];
      snprintf(tag, sizeof(tag), "%s_x%d", cat_name[g], nplans[g]);
      g_profiler.begin(s, 1, flops[g], abytes[g], tag, total_wg[g]);
    }
    if (g == 0) launch_T<1, 1>(d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else if (g == 1) launch_T<9, 9>(d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else if (g == 2) launch_T<16, 8>(d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else if (g == 3) launch_dma(wgrad_wave_dma_kernel, d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else if (g == 4) launch_dma(wgrad_band_dma_kernel<9, 9>, d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else if (g == 5) launch_dma(wgrad_band_dma_kernel<16, 8>, d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else if (g == 6) launch_dma(wgrad_direct_kernel<0>, d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s, 256);
    else if (g == 7) launch_dma(wgrad_direct_kernel<1>, d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s, 256);
    else if (g == 8) launch_dma(wgrad_direct_kernel<2>, d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s, 256);
    else launch_dma(wgrad_1x1_kernel, d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s, 256);
    if (fold_wgs[g]) {
      static const bool abl_fold = (dbm_abl_skip() & 8) != 0;  // (libdbm_measure.so only)
      if (pair_mode[g]) {
        if (!abl_fold)
          hipLaunchKernelGGL(wgrad_pair_fold_kernel, dim3(fold_wgs[g]), dim3(256), 0, s, d_plans[g], d_starts[g] + nplans[g] + 1, nplans[g]);
      } else
        hipLaunchKernelGGL(wgrad_fold_kernel, dim3(fold_wgs[g]), dim3(256), 0, s, d_plans[g], d_starts[g] + nplans[g] + 1, nplans[g]);
      DBM_HIP(hipGetLastError());
    }
    if (g_profiler.enabled) g_profiler.end(s);
  }
}

// single-layer form (op-level entry points / tests)
void launch_wgrad(const WgradDesc& d, hipStream_t s) {
  WgradBatch b;
  b.add(d);
  b.launch(s);
  DBM_HIP(hipStreamSynchronize(s));
  b.reset();
}
