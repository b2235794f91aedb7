// LDS-tiled implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) for the mid-size planes of the training step:
// the generator's full-resolution tail (srgan_train.py:556-574: post_upsample_conv_layer_1/2, the offset convolutions of the two
// deformable layers) and the discriminator's conv_layer1 .. conv_layer4 (srgan_train.py:626-634), forward AND data gradient.
//
// Why a second form next to igemm.hip (round 5, VERDICT r4 "next" #3): igemm_conv_kernel gathers BOTH MFMA operands from global
// memory -- one dword per lane and MFMA each -- and splits K over the wavefronts of a 32 x 32 tile; on these shapes it sits at
// 0.24-0.46 of the fp32 MFMA roof with 2.6x its algorithmic bytes in HBM traffic (every output-channel tile re-gathers its
// activations, tap by tap).  Here a workgroup (4 wavefronts) owns a BAND of output rows of ONE image and one 32-channel output tile:
//   * the band's input patch (all rows / columns any tap of any band position reaches, in LOGICAL coordinates -- a folded nearest x2
//     resize is resolved while staging --, zero-framed) and the matching slice of the packed weight image land in LDS by LDS-DMA
//     (global_load_lds), KC input channels at a time, double buffered: an activation is read from memory ONCE per workgroup instead
//     of once per tap, and a tap is an immediate offset on a ds_read_b32;
//   * no split-K: wavefront w owns position tiles w, w + 4, ... of the band over the WHOLE K (NT accumulator tiles per wavefront,
//     every A operand -- one ds_read_b32 of the weight slice -- feeds NT MFMAs); no cross-wavefront reduction, one barrier per chunk;
//   * the accumulators go out from the registers (lanes = consecutive positions: 128-byte runs) through the same epilogue as
//     igemm.hip's (bias, per-channel scale, residual axpys, accumulate, LeakyReLU, gradient mask).
// Numerics: exact fp32 products, fp32 accumulation over (channel chunk, channel pair, kernel row, kernel column) in that order.
#include "dbm_internal.h"
#include <cstdint>
#include <cstdlib>

namespace conv_tile {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// T taps (9: 3x3 stride 1; 16: 4x4 stride 2), SIN = input stride, OW = output columns of the plane, R = output rows per band, NT = position tiles per wavefront, KC = input
// channels per LDS chunk, PW = floats per DMA piece of the patch (4: global_load_lds_dwordx4, rows of 16-byte multiples; 3: dwordx3).
// Patch layout in LDS: rows of LW = PW + OW floats -- one PAD piece, then the row's OW / PW data pieces --, zero rows above / below
// the image.  The pad piece is never written by a DMA (it stays zero): its last float is the row's left frame cell, and the NEXT
// row's first pad float is this row's right frame cell, so a tap is still one immediate offset.  (First version: a 38-float row
// moved by dword pieces -- 7 instructions per channel plane instead of 2, and the piece rate, not the bytes, is what LDS-DMA pays
// for: MI355X_MICROARCH.md "ldsdma-fill"; the kernel ran at the SUM of its MFMA and DMA times.)
template <int T, int SIN, int OW, int R, int NT, int KC, int PW>
struct Geo {
  static constexpr int KW = T == 9 ? 3 : 4;          // 3x3 (pad 1, stride 1) or 4x4 (pad 1, stride 2) windows
  static constexpr int IW = (OW - 1) * SIN + KW - 2; // input columns a band reads, frame excluded: OW (3x3) / 2 OW (4x4 stride 2)
  static constexpr int LW = IW + PW;                 // patch row (floats): pad piece + data
  static constexpr int PPR = LW / PW;                // pieces per row (pad piece included)
  static constexpr int PR = (R - 1) * SIN + KW;      // patch rows
  static constexpr int CELLS = PR * LW;
  static constexpr int NP = PR * PPR;                // pieces per channel plane
  static constexpr int G = (NP + 63) / 64;           // DMA instructions per channel plane
  static constexpr int CS = (CELLS + 1 + 63) / 64 * 64;  // plane stride (floats; + the frame cell behind the last row)
  static constexpr int NPOS = R * OW;                // positions of a band
  static constexpr int NTILES = (NPOS + 31) / 32;
  static constexpr int WCH = T * KC * 32;            // weight floats of a chunk: [tap][channel][32 output channels]
  static constexpr int BUF = KC * CS + WCH;          // floats per buffer
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF * sizeof(float);
  static_assert((T == 9 && SIN == 1) || (T == 16 && SIN == 2), "3x3 stride 1 or 4x4 stride 2");
  static_assert(IW % PW == 0 && LW % PW == 0, "rows are whole pieces");
  static_assert(NTILES <= 4 * NT && NTILES > 4 * (NT - 1), "band does not fit the wavefronts' tiles");
  static_assert(KC % 2 == 0 && (T * KC) % 8 == 0, "a weight DMA instruction moves eight 32-float rows");
};

template <int V> struct NTsel { static constexpr int value = V; };
struct B0 { static constexpr int value = 0; };
struct B1 { static constexpr int value = 1; };

template <int T, int SIN, int OW, int R, int NT, int KC, int PW>
__global__ __launch_bounds__(256, 2) void conv_tile_kernel(const ConvDesc d) {
  using g = Geo<T, SIN, OW, R, NT, KC, PW>;
  constexpr int KW = g::KW, LW = g::LW, CS = g::CS, G = g::G;
  // TWO distinct LDS objects, not two halves of one array: hipcc's wait-count pass orders every LDS read behind every LDS-DMA in flight
  // that MAY alias it (s_waitcnt vmcnt(0) -- the DMA of the next chunk would never overlap this chunk's MFMAs); run-time offsets into
  // one array always may, reads of one object and a DMA into another provably do not
  __shared__ __attribute__((aligned(16))) float bufA[g::BUF];
  __shared__ __attribute__((aligned(16))) float bufB[g::BUF];
  float* const pA = bufA;   // (the lambdas below name the objects through these: clang's host pass does not instantiate a kernel template
  float* const pB = bufB;   //  whose nested generic lambdas refer to its static __shared__ locals directly -- the device stub goes missing)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int bands = (d.OHl + R - 1) / R;
  const int n = (int)blockIdx.x / bands, band = (int)blockIdx.x - n * bands;
  const int a0 = band * R;
  const int cout0 = (int)blockIdx.y * 32;
  const int dymin = d.dy[d.tmap[0]];   // canonical tap 0 = the window's top-left corner (-1, -1)

  // ---- staging tables: patch piece 64 q + lane -> source offset inside a channel plane (the same for every channel) ----
  int soff[G];
  unsigned valid = 0;
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const int e = 64 * q + lane;
    const int pr = e / g::PPR, pp = e - pr * g::PPR;
    const int iy = a0 * SIN + dymin + pr;   // LOGICAL input row (a folded nearest x2 resize -- PW == 1 only -- is resolved here)
    const bool ok = e < g::NP && pp >= 1 && (unsigned)iy < (unsigned)(d.Hin << d.ups);
    soff[q] = ok ? (iy >> d.ups) * d.Win + (((pp - 1) * PW) >> d.ups) : 0;
    valid |= ok ? (1u << q) : 0u;
    // pieces no DMA ever writes (the pad piece of every row, rows outside the image) are cleared once, in both buffers, by the wavefront
    // that owns the channel plane (wave, wave + 4, ...)
    if (!ok && e < g::NP) {
      for (int c = wave; c < KC; c += 4)
#pragma unroll
        for (int k = 0; k < PW; ++k) { bufA[c * CS + e * PW + k] = 0.f; bufB[c * CS + e * PW + k] = 0.f; }
    }
  }
  // (the float behind the last patch row is the right frame cell of that row's last column: inside the plane stride, never written)
  for (int c = wave; c < KC; c += 4)
    for (int e = g::CELLS + lane; e < CS; e += 64) { bufA[c * CS + e] = 0.f; bufB[c * CS + e] = 0.f; }
  const float* xn = d.x + (long)n * d.xsn;
  // weight slice: DMA instruction i moves rows 8 i .. 8 i + 7 of the chunk's [tap][channel] x 32 block (16 bytes per lane); this wavefront
  // issues instructions wave, wave + 4, ...  Their source offsets (without the chunk's first channel) are computed ONCE: d.tmap is
  // indexed with a run-time value -- a vector load of the kernel argument whose vmcnt(0) wait, inside the staging loop, stalled
  // every chunk's weight DMA behind the patch DMA in front of it (first version: 2x the MFMA-bound time).
  constexpr int WI = (T * KC / 8 + 3) / 4;
  long woff[WI];
#pragma unroll
  for (int k = 0; k < WI; ++k) {
    const int i = wave + 4 * k;
    const int r = 8 * i + (lane >> 3);                  // row of the chunk's block: tap r / KC, channel r % KC
    const int t = r < T * KC ? r / KC : 0, cl = r - (r / KC) * KC;
    woff[k] = ((long)d.tmap[t] * d.Cin + cl) * d.CoutP + cout0 + (lane & 7) * 4;
  }

  auto stage = [&](int chunk, auto BUFSEL) {
    float* smem = decltype(BUFSEL)::value ? pB : pA;
    const int c0 = chunk * KC;
#pragma unroll
    for (int k = 0; k < WI; ++k) {
      const int i = wave + 4 * k;
      // (the 16-byte form exists on gfx950 only: clang's HOST pass rejects the immediate, silently -- inside a generic lambda the
      //  error is a substitution failure and the kernel's device stub is simply never emitted; hence the guard)
#if defined(__HIP_DEVICE_COMPILE__)
      if (i < T * KC / 8) __builtin_amdgcn_global_load_lds(d.wp + woff[k] + (long)c0 * d.CoutP, (lds_ptr)(smem + KC * CS + i * 256), 16, 0, 0);
#endif
    }
    for (int c = wave; c < KC; c += 4) {
      const float* src = xn + (long)(c0 + c) * d.xsc;
#pragma unroll
      for (int q = 0; q < G; ++q) {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((valid >> q) & 1u) __builtin_amdgcn_global_load_lds(src + soff[q], (lds_ptr)(smem + c * CS + 64 * q * PW), 4 * PW, 0, 0);
#endif
      }
    }
  };

  // ---- this wavefront's position tiles: w, w + 4, ...; NTW of them (NT or NT - 1: a compile-time constant of the body below, which
  // is instantiated for both -- a run-time tile count would put an exec-mask branch around every MFMA) ----
  auto body = [&](auto NTW_) {
    constexpr int NTW = decltype(NTW_)::value;
    int cell[NTW];     // patch cell of the lane's position (tap (0, 0)), its channel of the pair included
    int pos[NTW];      // position inside the band (>= NPOS: padding lane)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int p = (wave + 4 * i) * 32 + j;
      pos[i] = p;
      const int pc = p < g::NPOS ? p : 0;
      const int al = pc / OW, b = pc - al * OW;
      cell[i] = al * SIN * LW + b * SIN + (PW - 1) + kh * CS;
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // One unit = one kernel row of one channel pair: KW weight reads + NTW * KW patch reads, NTW * KW MFMAs.  The reads of unit u + 1
    // are issued before the MFMAs of unit u (two register sets in ping-pong, pinned by scheduling barriers).
    constexpr int KH = KW;
    constexpr int UNITS = (KC / 2) * KH;
    auto load_unit = [&](int u, auto BUFSEL, float (&a)[KW], float (&b)[NTW][KW]) {
      const float* smem = decltype(BUFSEL)::value ? pB : pA;
      constexpr int base = 0;
      const int cp = u / KH, ky = u - cp * KH;
      const float* W = smem + base + KC * CS + (ky * KW * KC + 2 * cp) * 32 + lane;   // [tap][channel][32]: lane = kh * 32 + j
      const float* P = smem + base + 2 * cp * CS + ky * LW;
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) a[kx] = W[kx * KC * 32];
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) b[i][kx] = P[cell[i] + kx];
    };
    // (hipcc waits for LDS reads with lgkmcnt(0) only -- in this kernel it never counts -- so the reads of the NEXT unit are issued
    //  behind the FIRST MFMA of the current one, whose wait then covers nothing younger than its own operands, and complete in the
    //  shadow of the unit's remaining MFMAs)
    auto mma_unit = [&](const float (&a)[KW], const float (&b)[NTW][KW], bool first) {
#pragma unroll
      for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int i = 0; i < NTW; ++i)
          if ((kx == 0 && i == 0) == first) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kx], b[i][kx], acc[i], 0, 0, 0);
    };
    auto compute = [&](auto BUFSEL) {
      float a0r[KW], b0r[NTW][KW], a1r[KW], b1r[NTW][KW];
      load_unit(0, BUFSEL, a0r, b0r);
#pragma unroll
      for (int u = 0; u < UNITS; u += 2) {
        mma_unit(a0r, b0r, true);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < UNITS) load_unit(u + 1, BUFSEL, a1r, b1r);
        __builtin_amdgcn_sched_barrier(0);
        mma_unit(a0r, b0r, false);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < UNITS) {
          mma_unit(a1r, b1r, true);
          __builtin_amdgcn_sched_barrier(0);
          if (u + 2 < UNITS) load_unit(u + 2, BUFSEL, a0r, b0r);
          __builtin_amdgcn_sched_barrier(0);
          mma_unit(a1r, b1r, false);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    const int nchunk = ((d.cin_live > 0 && d.cin_live < d.Cin) ? (d.cin_live + KC - 1) / KC * KC : d.Cin) / KC;
    __syncthreads();   // (the frame cells are cleared before any DMA may land beside them -- and before anybody reads)
    stage(0, B0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // chunk c is computed from buffer c & 1 while chunk c + 1 lands in the other one (two chunks per trip: the buffer offsets are
    // compile-time constants, so that hipcc can tell the DMA's destination from the reads' source -- conv_cl16.hip)
    for (int c = 0; c < nchunk; c += 2) {
      if (c + 1 < nchunk && !DBM_ABL_BIT(d, 1)) stage(c + 1, B1{});
      if (!DBM_ABL_BIT(d, 2)) compute(B0{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (c + 1 >= nchunk) break;
      if (c + 2 < nchunk && !DBM_ABL_BIT(d, 1)) stage(c + 2, B0{});
      if (!DBM_ABL_BIT(d, 2)) compute(B1{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }

    // ---- epilogue: the accumulators go out from the registers (lanes j = consecutive positions: 128-byte runs) ----
    // vmcnt counts stores as well as loads, in order: a load issued BEHIND a store cannot be awaited without draining that store
    // (a full write round trip, ~2 us).  The first version loaded its operands four channels at a time between the stores -- twelve
    // drains per wavefront, 25 of the kernel's 88 us.  Now: what depends on the channel only (bias, scale) is loaded once; what depends on
    // the position (residuals, accumulate target, mask) is loaded for a WHOLE tile, and for tile i + 1 BEFORE tile i's stores.
    if (DBM_ABL_BIT(d, 4)) {
#pragma unroll
      for (int i = 0; i < NTW; ++i)
        if (acc[i][0] == 12345.f) d.y[0] = acc[i][1];
      return;
    }
    // Raw buffer accesses (bounds-checked by the hardware): an offset of -1 drops a store / reads zero, so padding lanes and
    // channels past Cout need no branch -- behind a branch hipcc forgets its vmcnt bookkeeping and waits with vmcnt(0) -- and an
    // absent operand is a zero-length buffer.
    const int cmax = d.Cout - 1;
    auto rsrc = [](const void* ptr, long bytes) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, ptr ? (int)bytes : 0, 0x00020000);
    };
    const long ybytes = 4L * ((long)(d.N - 1) * d.ysn + (long)d.Cout * d.ysc);
    const __amdgpu_buffer_rsrc_t ry = rsrc(d.y, ybytes);
    const __amdgpu_buffer_rsrc_t rb = rsrc(d.bias, 4L * d.Cout), rsc = rsrc(d.ch_scale, 4L * d.Cout);
    auto ldf = [](__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); };
    float e_b[16], e_s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = cout0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const int off = c <= cmax ? 4 * c : -1;
      e_b[r] = ldf(rb, off);
      const float sc = ldf(rsc, off);
      e_s[r] = d.ch_scale ? sc : 1.f;
    }
    // byte offset of (image n, channel cout0 + 4 kh, the lane's position of tile i) inside y; -1: nothing to store
    auto tile_off = [&](int i) {
      const int p = pos[i];
      const int pq = p < g::NPOS ? p : 0;
      const int al = pq / OW, b = pq - al * OW;
      const int a = a0 + al;
      const bool ok = p < g::NPOS && a < d.OHl;
      const long e = (long)n * d.ysn + (long)(cout0 + 4 * kh) * d.ysc + (long)(a * d.so + d.oy0) * d.OWp + (b * d.so + d.ox0);
      return ok ? (int)(4 * e) : -1;
    };
    auto reg_off = [&](int base, int r) {   // register r of a tile: channel + (r & 3) + 8 (r >> 2)
      const int c = cout0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      return (base >= 0 && c <= cmax) ? base + 4 * ((r & 3) + 8 * (r >> 2)) * d.ysc : -1;
    };
    const bool extra = d.r1 || d.r2 || d.accumulate || d.mask;   // (uniform: one of two straight-line instances below)
    if (!extra) {
      // (optional channels-last twin of a 64-channel output -- the next layer's fused deformable sampler reads that layout: registers
      //  4 g .. 4 g + 3 of a lane are four consecutive channels: one 16-byte store; saves the nchw_to_nhwc64 pass behind this launch)
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t ryt = rsrc(d.yt, 256L * d.N * d.OHl * d.OWl);
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int base = tile_off(i);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = (acc[i][r] * e_s[r] + e_b[r]) * d.s1;
          if (d.act) v[r] = v[r] >= 0.f ? v[r] : d.slope * v[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ry, reg_off(base, r), 0, 0);
        }
        if (d.yt) {
          const int p = pos[i];
          const int pq = p < g::NPOS ? p : 0;
          const int al = pq / OW, b = pq - al * OW;
          const bool ok = p < g::NPOS && a0 + al < d.OHl;
          const int pixo = ok ? 4 * (((n * d.OHl + a0 + al) * d.OWl + b) * 64 + cout0 + 4 * kh) : -1;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            __builtin_amdgcn_raw_buffer_store_b128((u4){__builtin_bit_cast(unsigned, v[4 * q4]), __builtin_bit_cast(unsigned, v[4 * q4 + 1]),
                                                        __builtin_bit_cast(unsigned, v[4 * q4 + 2]), __builtin_bit_cast(unsigned, v[4 * q4 + 3])},
                                                   ryt, pixo >= 0 ? pixo + 32 * q4 : -1, 0, 0);
        }
      }
    } else {
      // position-dependent operands, eight registers (half a tile) at a time, the NEXT half's loads issued before this half's stores.
      // r1 / r2 / mask are addressed like y (same channel stride, same spatial index), with their own image strides.
      const __amdgpu_buffer_rsrc_t r1r = rsrc(d.r1, 4L * ((long)(d.N - 1) * d.r1sn + (long)d.Cout * d.ysc));
      const __amdgpu_buffer_rsrc_t r2r = rsrc(d.r2, 4L * ((long)(d.N - 1) * d.r2sn + (long)d.Cout * d.ysc));
      const __amdgpu_buffer_rsrc_t rmk = rsrc(d.mask, 4L * ((long)(d.N - 1) * d.masksn + (long)d.Cout * d.ysc));
      const __amdgpu_buffer_rsrc_t rya = rsrc(d.accumulate ? d.y : nullptr, ybytes);
      const int dn1 = (int)(4 * (long)n * (d.r1sn - d.ysn)), dn2 = (int)(4 * (long)n * (d.r2sn - d.ysn)), dnm = (int)(4 * (long)n * (d.masksn - d.ysn));
      float o_r1[2][8], o_r2[2][8], o_y[2][8], o_m[2][8];
      auto load_ops = [&](int h, float (&q1)[8], float (&q2)[8], float (&qy)[8], float (&qm)[8]) {
        const int base = tile_off(h >> 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = 8 * (h & 1) + k;
          const int c = cout0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int off = reg_off(base, r);
          q1[k] = ldf(r1r, (off >= 0 && c < d.r1_nch) ? off + dn1 : -1);
          q2[k] = ldf(r2r, off >= 0 ? off + dn2 : -1);
          qy[k] = ldf(rya, off);
          qm[k] = ldf(rmk, (off >= 0 && c >= d.mask_c0) ? off + dnm : -1);
        }
      };
      load_ops(0, o_r1[0], o_r2[0], o_y[0], o_m[0]);
#pragma unroll
      for (int h = 0; h < 2 * NTW; ++h) {
        if (h + 1 < 2 * NTW) load_ops(h + 1, o_r1[(h + 1) & 1], o_r2[(h + 1) & 1], o_y[(h + 1) & 1], o_m[(h + 1) & 1]);
        const int base = tile_off(h >> 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = 8 * (h & 1) + k;
          float v = (acc[h >> 1][r] * e_s[r] + e_b[r]) * d.s1 + d.r1s * o_r1[h & 1][k];
          if (d.r2) v = d.s2 * v + o_r2[h & 1][k];
          v += o_y[h & 1][k];
          if (d.act) v = v >= 0.f ? v : d.slope * v;
          v = o_m[h & 1][k] >= 0.f ? v : d.slope * v;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, reg_off(base, r), 0, 0);
        }
      }
    }
  };
  // (every wavefront passes the same number of barriers whichever instance it runs)
  const int wu = __builtin_amdgcn_readfirstlane(wave);
  if constexpr (NT == 1) {
    body(NTsel<1>{});   // (a wavefront without a tile multiplies padding lanes: it has to pass the barriers anyway)
  } else {
    if (wu + 4 * (NT - 1) < g::NTILES) body(NTsel<NT>{});
    else body(NTsel<NT - 1>{});
  }
}

template <int T, int SIN, int OW, int R, int NT, int KC, int PW>
static void launch_cfg(const ConvDesc& d, hipStream_t s) {
  using g = Geo<T, SIN, OW, R, NT, KC, PW>;
  static_assert(g::LDS_BYTES <= 64 * 1024, "static LDS");
  const int bands = (d.OHl + R - 1) / R;
  dim3 grid((unsigned)(d.N * bands), (unsigned)((d.Cout + 31) / 32), 1u);
  hipLaunchKernelGGL((conv_tile_kernel<T, SIN, OW, R, NT, KC, PW>), grid, dim3(256), 0, s, d);
}

}  // namespace conv_tile
using namespace conv_tile;

// Fills d.tmap (canonical tap u = (ky, kx) row-major over the window -> index into d.dy / d.dx) when the taps are exactly one full
// K x K window; false otherwise.
static bool canonical_taps(ConvDesc& d, int K) {
  int dymin = 127, dxmin = 127;
  for (int t = 0; t < d.T; ++t) { dymin = d.dy[t] < dymin ? d.dy[t] : dymin; dxmin = d.dx[t] < dxmin ? d.dx[t] : dxmin; }
  bool seen[DBM_MAX_TAPS] = {};
  for (int t = 0; t < d.T; ++t) {
    const int ky = d.dy[t] - dymin, kx = d.dx[t] - dxmin;
    if (ky < 0 || ky >= K || kx < 0 || kx >= K || seen[ky * K + kx]) return false;
    seen[ky * K + kx] = true;
    d.tmap[ky * K + kx] = (signed char)t;
  }
  return true;
}

// Plans the LDS-tiled form for a launch: 0 = not served (the caller falls back to igemm_conv_kernel), else the configuration id
// (> 0); fills d.tmap and *wgs (workgroups of the launch).  DBM_CONV_TILE=0 switches the form off (A/B against igemm.hip).
bool conv_tile_writes_yt(const ConvDesc& d_in) {
  ConvDesc d = d_in;
  long wgs = 0;
  static const int yt_env = getenv("DBM_CONV_TILE_YT") ? atoi(getenv("DBM_CONV_TILE_YT")) : 1;   // (A/B switch)
  return yt_env && d.yt != nullptr && conv_tile_plan(d, &wgs) != 0;
}

int conv_tile_plan(ConvDesc& d, long* wgs) {
  static const int enable = getenv("DBM_CONV_TILE") ? atoi(getenv("DBM_CONV_TILE")) : 1;
  if (!enable) return 0;
  if (d.wp16 || d.nphase > 1 || d.Cin % 8 != 0 || d.CoutP % 32 != 0 || d.N < 1) return 0;
#ifdef DBM_MEASURE
  d.abl = DBM_MEASURE_ENV("CT_ABL");
#endif
  if (d.yt && (d.Cout != 64 || d.so != 1 || d.r1 || d.r2 || d.accumulate || d.mask || 256L * d.N * d.OHl * d.OWl >= (1L << 31))) return 0;
  const bool k3 = d.T == 9 && d.sin == 1, k4 = d.T == 16 && d.sin == 2 && d.ups == 0;
  if (!(k3 || k4) || d.OHl != d.OWl) return 0;
  const int Hl = d.Hin << d.ups, Wl = d.Win << d.ups;
  if (k3 && !(Hl == d.OHl && Wl == d.OWl)) return 0;
  if (k4 && !(Hl == 2 * d.OHl && Wl == 2 * d.OWl)) return 0;
  if (!canonical_taps(d, k3 ? 3 : 4) || d.dy[d.tmap[0]] != -1 || d.dx[d.tmap[0]] != -1) return 0;   // (pad 1)
  {  // the epilogue addresses y / r1 / r2 / mask through 32-bit byte offsets (raw buffer accesses)
    const long lim = (1L << 31) / 4 - 1, per = (long)d.Cout * d.ysc;
    if ((long)(d.N - 1) * d.ysn + per > lim || (d.r1 && (long)(d.N - 1) * d.r1sn + per > lim) || (d.r2 && (long)(d.N - 1) * d.r2sn + per > lim) ||
        (d.mask && (long)(d.N - 1) * d.masksn + per > lim))
      return 0;
  }
  const long mt = (d.Cout + 31) / 32;
  const bool x4 = !d.ups && (reinterpret_cast<uintptr_t>(d.x) & 15) == 0 && (d.xsn & 3) == 0;   // 16-byte DMA pieces
  // Which plane classes take this form.  Measured INSIDE the training step (tools/experiments/ab_env.sh, two alternations): the 36 x 36
  // layers (generator tail: the iteration's critical path) and the 18 x 18 ones pay -- 7.92 against 7.98 ms.  The 4x4 stride-2 layers
  // (51.6 -> 43.8 us, 56.3 -> 47.8 us standalone) cost + 0.04 ms per step when first measured and nothing on the final round-5 schedule
  // (7.66 / 7.68 against 7.67 / 7.69): on.  The 9 x 9 planes (34.4 -> 30.7 us standalone) cost + 0.02 then and + 0.23 ms now: these
  // discriminator layers run on the 64 CUs a persistent trunk launch leaves, where igemm_conv_kernel's small workgroups (up to eight
  // per CU) use a CU better than 256 workgroups of four wavefronts.  Off; DBM_CONV_TILE_K4=0 / DBM_CONV_TILE_9=1 for the A/B.
  static const int k4_enable = getenv("DBM_CONV_TILE_K4") ? atoi(getenv("DBM_CONV_TILE_K4")) : 1;
  static const int p9_enable = getenv("DBM_CONV_TILE_9") ? atoi(getenv("DBM_CONV_TILE_9")) : 0;
  if (k4) {
    if (!k4_enable) return 0;
    if (d.OWl == 18 && x4) {   // 36 x 36 -> 18 x 18 (discriminator conv_layer1): two bands of nine output rows, six tiles each
      *wgs = (long)d.N * 2 * mt;
      return 5;
    }
    if (d.OWl == 9) {          // 18 x 18 -> 9 x 9 (conv_layer3): the whole image, three tiles
      *wgs = (long)d.N * mt;
      return 6;
    }
    return 0;
  }
  if (d.OWl == 36) {   // four bands of nine rows: 324 positions = 11 tiles per workgroup
    *wgs = (long)d.N * 4 * mt;
    return x4 ? 1 : 4;
  }
  if (d.OWl == 18) {   // (inside the step: neutral, 7.94-7.96 against 7.95-7.96 without this class -- profiles/r5/ab_conv_tile_18.txt)
    // the whole image per workgroup (324 positions, 11 tiles) when that still makes >= 256 workgroups; else two bands of nine rows
    if ((long)d.N * mt >= 256) {
      *wgs = (long)d.N * mt;
      return 2;
    }
    *wgs = (long)d.N * 2 * mt;
    return 3;
  }
  if (d.OWl == 9 && p9_enable && (long)d.N * mt >= 256) {   // 9 x 9 planes (discriminator conv_layer4): one image = three tiles
    // (fewer workgroups than CUs: igemm_conv_kernel's K split over 8 / 16 wavefronts is faster -- 23.5 vs 29.6 us at 128)
    *wgs = (long)d.N * mt;
    return 7;
  }
  return 0;
}

void conv_tile_launch(const ConvDesc& d, int cfg, hipStream_t s) {
  switch (cfg) {
    case 1: launch_cfg<9, 1, 36, 9, 3, 8, 4>(d, s); break;
    // (dword pieces for the 72-byte rows.  Round 5's 12-byte form gave wrong results -- root cause, round 6, tools/experiments/ubench/
    //  lds_dma_b96.hip: global_load_lds_dwordx3 lands lane l's three dwords at dst + 16 l, a 16-BYTE lane stride with every fourth dword
    //  untouched, not at dst + 12 l -- the staging tables here assume lane-linear pieces of PW dwords.  profiles/r6/lds_dma_b96.txt)
    case 2: launch_cfg<9, 1, 18, 18, 3, 8, 1>(d, s); break;
    case 3: launch_cfg<9, 1, 18, 9, 2, 8, 1>(d, s); break;
    case 4: launch_cfg<9, 1, 36, 9, 3, 8, 1>(d, s); break;    // (a folded x2 resize, or rows that are not 16-byte aligned)
    case 5: launch_cfg<16, 2, 18, 9, 2, 4, 4>(d, s); break;   // 4x4 stride 2 from 36-wide rows (16-byte pieces), four channels per chunk
    case 6: launch_cfg<16, 2, 9, 9, 1, 8, 1>(d, s); break;    // 4x4 stride 2 from 18-wide rows
    case 7: launch_cfg<9, 1, 9, 9, 1, 8, 1>(d, s); break;     // 3x3 on 9 x 9 planes
    default: DBM_CHECK(false, "conv_tile_launch: unknown configuration");
  }
  DBM_HIP(hipGetLastError());
}
