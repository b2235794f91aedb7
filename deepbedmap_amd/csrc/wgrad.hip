// Weight-gradient of the convolutions (backward of the L.Convolution2D call sites listed in
// igemm.hip) on the fp32 matrix cores:  gW[o][c][t] += scale * sum_{n,a,b} dy[n][o][a][b] * x[n][c][tap t].
//
// GEMM view: M = out channels, N = (in channel, tap), K = output positions.  The contiguous
// memory axis (positions) is the MFMA K axis, which per-lane global loads cannot feed
// efficiently (every lane would touch its own cache line), so operands are staged through LDS:
// a workgroup stages, for one "band" (IB images x R output rows), the 32 x BP slab of dy and
// the zero-padded, tap-ready input patch of up to 128 input channels, then each wavefront owns
// one 32-channel input tile and keeps T (= taps) independent 32x32 accumulators, i.e. T
// independent MFMA chains fed by 1 + 2/T LDS dwords per MFMA.
//
// One layer alone cannot fill 256 CUs (the trunk's weight matrices have 2..12 32x32 tiles), and
// splitting K harder only multiplies the fp32 atomics that fold the partial sums.  So the launch
// is BATCHED: a device-resident table of per-layer plans, one workgroup = (layer, input-channel
// group, output tile, K split); all weight gradients of a backward pass with the same kernel size
// go out in a single launch after the data-gradient chain (they do not depend on each other).
#include "dbm_internal.h"
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
      // K loop, software pipelined by one step: the A value and the patch offset of step k+1 are fetched from LDS
      // while the TPW MFMAs of step k issue (the B reads of a step depend on its offset).
      const float* arow = ldsY + j * p.YS + kh;
      const float* xrow = ldsX + (ct * 32 + j) * p.XS;
      float av = arow[0];
      int off = pixoff[kh];
      for (int kp = 0; kp < p.BPp; kp += 2) {
        const float* xb = xrow + off;
        float bv[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) bv[t] = xb[toff[t]];
        const int kn = (kp + 2 < p.BPp) ? kp + 2 : kp;
        const float av_n = arow[kn];
        const int off_n = pixoff[kn + kh];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
        av = av_n;
        off = off_n;
      }
    }
  }

  // ---- fold the partial sums into gW[o][c][t] (t fastest).  A lane owns (o, c) pairs with a stride of T floats
  // between lanes, so direct atomics would touch a different cache line per lane; instead the wavefronts of one
  // input tile transpose 8 output rows at a time through LDS and issue the atomics over consecutive addresses. ----
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
          if (o < d.Cout && c < d.Cin) atomicAdd(d.gW + ((long)o * d.Cin + cin_w) * T + rem, tw[e]);
        }
      }
      __syncthreads();
    }
  }
  if (d.gb && bx == 0 && tid < 256 && (tid & 7) == 0 && cout0 + (tid >> 3) < d.Cout)
    atomicAdd(d.gb + cout0 + (tid >> 3), d.scale * bsum);
}

static inline int odd_up(int v) { return v | 1; }

size_t wgrad_plan(const WgradDesc& d, WgradPlan& p, int level) {
  const int T = d.KH * d.KW;
  DBM_CHECK(T == 1 || T == 9 || T == 16, "wgrad: supported kernels are 1x1, 3x3, 4x4");
  DBM_CHECK(d.OW < 4096 && d.OH < 4096, "wgrad: image too large");
  p.d = d;
  const int tiles = (d.Cin + 31) / 32;
  p.groups = (tiles + 3) / 4;
  p.G = (tiles + p.groups - 1) / p.groups;
  p.coutTiles = (d.Cout + 31) / 32;
  // band selection: whole images if small, else row bands of one image; keep LDS <= ~100 KB
  const int Wst = (d.OW - 1) * d.stride + d.KW;
  const long budget = (T <= 9) ? 19800 : 25000;  // floats; <= 79 KB lets two workgroups share a CU's 160 KB
  auto cost = [&](int IB, int R) {
    const int Rin = (R - 1) * d.stride + d.KH;
    const long bpp = (IB * R * d.OW + 1) & ~1;
    return (long)p.G * 32 * odd_up(IB * Rin * Wst) + 32L * odd_up((int)bpp) + 2 * bpp + (long)IB * Rin * Wst;
  };
  int IB = 1, R = d.OH;
  if (cost(1, d.OH) <= budget) {
    while (IB * 2 <= d.N && IB * 2 <= 128 && IB * 2 * d.OH * d.OW <= (324 >> level) && cost(IB * 2, d.OH) <= budget) IB *= 2;
  } else {
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
  const long positions = (long)d.N * d.OH * d.OW;
  // (~8 images of the 9x9 trunk per workgroup; `level` > 0 halves that, and the band, per step: used by batches
  // that would otherwise leave most of the chip idle)
  const long per_wg = std::max(1L, 648L >> level);
  int S = (int)((positions + per_wg - 1) / per_wg);
  if (S > p.nbands) S = p.nbands;
  if (S < 1) S = 1;
  p.S = S;
  p.wg_count = p.groups * p.coutTiles * S;
  auto magic = [](unsigned dv) { return (unsigned)((0x100000000ULL + dv - 1) / dv); };
  const int plane = d.Hin * d.Win;
  p.planeM = magic((unsigned)plane);
  p.winM = magic((unsigned)d.Win);
  const int tail_ci = d.Cin - (p.groups - 1) * p.G * 32;  // channels of the last input group
  p.fast = p.nbr == 1 && d.ups == 0 && d.xsc == plane && d.dysc == d.OH * d.OW && (d.xsn % 4) == 0 && (d.dysn % 4) == 0 &&
           ((uintptr_t)d.x % 16) == 0 && ((uintptr_t)d.dy % 16) == 0 && ((32L * d.dysc) % 4) == 0 &&
           (((long)p.G * 32 * d.xsc) % 4) == 0 && ((long)std::min(32, d.Cout) * d.OH * d.OW) % 4 == 0 &&
           ((long)(d.Cout % 32 ? d.Cout % 32 : 32) * d.OH * d.OW) % 4 == 0 && ((long)std::min(p.G * 32, tail_ci) * plane) % 4 == 0 &&
           ((long)std::min(p.G * 32, d.Cin) * plane) % 4 == 0 && (long)p.G * 32 * plane < (1L << 22);
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
  for (int g = 0; g < 3; ++g) {
    if (d_plans[g]) (void)hipFree(d_plans[g]);
    if (d_starts[g]) (void)hipFree(d_starts[g]);
    d_plans[g] = nullptr;
    d_starts[g] = nullptr;
    nplans[g] = total_wg[g] = 0;
    lds[g] = 0;
    flops[g] = 0.0;
  }
}

void WgradBatch::build() {
  static const int TT[3] = {1, 9, 16};
  for (int g = 0; g < 3; ++g) {
    std::vector<WgradPlan> plans;
    std::vector<int> starts;
    int total = 0;
    size_t maxlds = 0;
    double fl = 0.0;
    // a launch should offer about two workgroups per CU; small batches split their position axis finer
    for (int level = 0; level < 4; ++level) {
      plans.clear(); starts.clear();
      total = 0; maxlds = 0; fl = 0.0;
      for (const auto& d : descs) {
        if (d.KH * d.KW != TT[g]) continue;
        WgradPlan p;
        maxlds = std::max(maxlds, wgrad_plan(d, p, level));
        starts.push_back(total);
        total += p.wg_count;
        plans.push_back(p);
        fl += 2.0 * (double)d.N * d.OH * d.OW * d.Cout * d.Cin * TT[g];
      }
      if (total >= 448 || plans.empty()) break;
    }
    starts.push_back(total);
    nplans[g] = (int)plans.size();
    total_wg[g] = total;
    lds[g] = maxlds;
    flops[g] = fl;
    if (plans.empty()) continue;
    DBM_HIP(hipMalloc((void**)&d_plans[g], plans.size() * sizeof(WgradPlan)));
    DBM_HIP(hipMalloc((void**)&d_starts[g], starts.size() * sizeof(int)));
    DBM_HIP(hipMemcpy(d_plans[g], plans.data(), plans.size() * sizeof(WgradPlan), hipMemcpyHostToDevice));
    DBM_HIP(hipMemcpy(d_starts[g], starts.data(), starts.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  DBM_HIP(hipDeviceSynchronize());
  built = true;
}

void WgradBatch::launch(hipStream_t s) {
  if (!built) build();
  for (int g = 0; g < 3; ++g) {
    if (nplans[g] == 0) continue;
    if (g_profiler.enabled) g_profiler.begin(s, 1, flops[g]);
    if (g == 0) launch_T<1, 1>(d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else if (g == 1) launch_T<9, 9>(d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
    else launch_T<16, 8>(d_plans[g], d_starts[g], nplans[g], total_wg[g], lds[g], s);
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
