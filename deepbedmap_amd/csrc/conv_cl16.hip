// 3x3 / stride 1 / pad 1 convolution on CHANNELS-LAST bf16 activations, LDS-tiled, v_mfma_f32_32x32x16_bf16: the RRDB trunk
// (srgan_train.py:333-360, 393-404, 546) of the bf16 area sweep (BASELINE config 5; deepbedmap.py:689-741) on planes too
// large for the persistent 9x9 kernels -- a 288 x 288 crop has a 286 x 286 trunk plane, 81 796 pixels.
//
// Why not the per-layer implicit GEMM of igemm.hip in its bf16 form: it gathers fp32 NCHW activations straight from L1
// into MFMA operands (one dword per lane, channel and tap) and rounds them in registers.  At the bf16 MFMA rate (32 cycles
// per 32 x 32 x 16) the vector L1 then bounds the kernel at 5 % of the matrix peak.  Here
//   * activations live in HBM as NHWC bf16 (the dense block's concat: 192 channels = 384 bytes per pixel; the 64-channel
//     residual stream additionally as NHWC fp32, so that the 36 residual additions never round to 8 bits): a lane's eight K
//     values of the B operand are ONE 16-byte LDS read, for every tap, and a layer's output is written once, in bf16;
//   * a workgroup owns a tile of 16 columns x 2 * nslots rows (nslots <= 12: chosen by the launcher so that the plane is one
//     round of <= 256 workgroups) and stages, per chunk of 32 input channels, the tile's zero-framed halo block
//     ((TH + 2) x 18 pixels x 64 bytes) AND the chunk's weights (9 taps x 32 x Cout, in fragment order) into LDS, double
//     buffered: one barrier per chunk; the loads of chunk c + 1 are issued before the MFMAs of chunk c and written after them;
//   * both land by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass (measured: the register-staged
//     first version spent a quarter of a chunk's cycles in ds_write_b128);
//   * the MFMA's N axis (32 lanes) is a patch of 2 rows x 16 columns, dealt to the lanes so that every 16-lane group of a
//     ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) reads sixteen CONSECUTIVE pixels
//     of one row; a pixel's four 16-byte parts sit at slot (part ^ ((pixel >> 2) & 3)) of its 64 bytes, so sixteen
//     consecutive pixels cover the sixteen slots of the 256-byte bank row: conflict-free for every tap, no padding;
//   * wave w owns patches w and w + 8: waves w and w + 4 share a SIMD, so a tile of 12 patches is three per SIMD; the weight
//     fragment of a K step is read once per wave and used for both patches (and for both 32-channel output tiles of
//     conv_layer5): 1.0-1.5 KB of LDS reads per MFMA against the 2 KB an unblocked wave would need (LDS: 256 B / clk).
// Epilogues: bias + LeakyReLU -> bf16 into the concat's next 32 channels (conv_layer1..4); conv_layer5: fp32
// `a5 * rs + a0` (and `... * rs + x` for every third block, :358 / :402) against the fp32 residual stream, written as fp32
// AND as the next block's bf16 channels 0..63.
// Numerics = the bf16 mode of igemm.hip: operands rounded to nearest-even bf16, fp32 accumulation, fp32 residual stream.
#include "dbm_internal.h"
#include "kernels.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int CL_TW = 16;            // tile width (pixels)
constexpr int CL_HW = CL_TW + 2;     // halo block width
constexpr int CL_NT = 512;           // threads per workgroup (8 wavefronts)
constexpr int CL_MAXSLOTS = 12;
constexpr int CL_ACT_BYTES = (((2 * CL_MAXSLOTS + 2) * CL_HW * 4 + 63) / 64) * 1024;  // one buffer of staged activations

struct ClConvArgs {
  const __bf16* x; int xc;           // input: NHWC bf16, xc channels per pixel; channels [0, Cin) are read
  int Cin;                           // multiple of 32
  const bf16x8* w;                   // packed weights [chunk][tap][k half][mtile][lane][8]  (launch_pack_cl16)
  const float* bias;                 // Cout floats
  __bf16* y16; int yc, y0;           // bf16 output: NHWC, yc channels per pixel, first output channel y0 (may be null)
  float* y32;                        // fp32 output: NHWC, 64 channels per pixel (may be null)
  const float* r1; float s1;         // v = s1 * (acc + bias) + r1   (r1: NHWC fp32, 64 channels; may be null)
  const float* r2; float s2;         // then v = s2 * v + r2         (may be null)
  int act; float slope;
  int N, H, W, nslots, tilesX, tilesY;
  const void* zeros;                 // >= 16 bytes of device zeros (out-of-plane pixels of the halo block)
#ifdef DBM_MEASURE
  int abl;                           // libdbm_measure.so only (results wrong): 1 no MFMA loop, 2 no epilogue, 4 no staging after chunk 0
#endif
};

// half-lane h (0..31) -> (row 0/1, column 0..15) of the wavefront's 2 x 16 patch: each 16-lane group of a ds_read_b128
// covers one row
__device__ __forceinline__ void patch_of(int h, int& g, int& i) {
  if (h < 4) { g = 0; i = h; }
  else if (h < 12) { g = 1; i = h - 4; }
  else if (h < 16) { g = 0; i = h - 8; }
  else if (h < 20) { g = 1; i = h - 8; }
  else if (h < 28) { g = 0; i = h - 12; }
  else { g = 1; i = h - 16; }
}

typedef __attribute__((address_space(3))) void* lds_ptr;

__host__ __device__ constexpr int cl_act_bytes(int maxs) { return (((2 * maxs + 2) * CL_HW * 4 + 63) / 64) * 1024; }

// TSLOTS: the largest tile height (in 2-row patches) the LDS layout is sized for.  12: one workgroup per CU (96 / 132 KB), what a
// single crop wants (234 tiles of eleven patches on 256 CUs); 8: 78 KB with one output-channel tile -- two workgroups per CU, so
// that one's staging and epilogue run under the other's MFMAs when a launch has several rounds of tiles (crops batched per forward).
template <int MT, int TSLOTS>
__global__ __launch_bounds__(CL_NT, TSLOTS <= 8 ? 2 : 1) void conv_cl16_kernel(ClConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int TH = 2 * a.nslots, HR = TH + 2, NPIX = HR * CL_HW;
  const int AINS = (NPIX * 4 + 63) >> 6;         // LDS-DMA wave-instructions (1 KB each) per chunk of activations
  constexpr int ACT_BYTES = cl_act_bytes(TSLOTS);   // (the layout is the same for every tile height up to TSLOTS: compile-time offsets)
  constexpr int WINS = 18 * MT;                  // ... and of weights
  constexpr int W_BYTES = WINS * 1024;
  // (byte offsets into smem, not pointers: a pointer picked from an array by a run-time index loses its LDS address space
  // and every access turns into a flat_load)
  constexpr int WGT0 = 2 * ACT_BYTES;

  int b = blockIdx.x;
  const int tx = b % a.tilesX; b /= a.tilesX;
  const int ty = b % a.tilesY;
  int n = b / a.tilesY;
  const int gy0 = ty * TH - 1, gx0 = tx * CL_TW - 1;   // plane coordinates of halo pixel (0, 0)
  long img = (long)n * a.H * a.W;

  // ---- staging by LDS-DMA (global_load_lds_dwordx4: per-lane source, lane-linear destination; no staging registers, no
  // ds_write pass).  Activations: 64 bytes per pixel, the four 16-byte parts of pixel q stored at slot (part ^ ((q >> 2) & 3))
  // -- sixteen consecutive pixels then cover the sixteen slots of the 256-byte bank row whatever the part: the fragment
  // reads below stay conflict-free without padding; the permutation is applied on the SOURCE side (lane -> part), the
  // destination stays linear.  Out-of-plane pixels are fetched from a zero block. ----
  // (a bound that depends on TSLOTS -- (ACT_BYTES / 1024 + 7) / 8 -- makes hipcc drop the host stub of every instantiation
  // without a diagnostic: the arrays below are captured by the generic lambdas)
  constexpr int ASTEPS = (cl_act_bytes(CL_MAXSLOTS) / 1024 + 7) / 8;   // AINS <= 30 instructions over eight wavefronts
  constexpr int WSTEPS = (WINS + 7) / 8;
  const __bf16* asrc[ASTEPS];
  const __bf16* zsrc = reinterpret_cast<const __bf16*>(a.zeros);
#pragma unroll
  for (int s = 0; s < ASTEPS; ++s) {
    const int u = 64 * (wave + 8 * s) + lane;
    const int q = u >> 2, part = (u & 3) ^ ((q >> 2) & 3);
    const int hy = q / CL_HW, hx = q - hy * CL_HW;
    const int gy = gy0 + hy, gx = gx0 + hx;
    const bool inside = q < NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    asrc[s] = inside ? a.x + (img + (long)gy * a.W + gx) * a.xc + 8 * part : nullptr;
  }
  const bf16x8* wcur = a.w;
  auto stage = [&](int c, auto BUF) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < ASTEPS; ++s) {
      const int k = wave + 8 * s;
      if (k < AINS)
        __builtin_amdgcn_global_load_lds(asrc[s] ? asrc[s] + 32 * c : zsrc, (lds_ptr)(smem + buf * ACT_BYTES + k * 1024), 16, 0, 0);
    }
    const u4v* wsrc = reinterpret_cast<const u4v*>(wcur) + (long)c * (WINS * 64) + lane;
#pragma unroll
    for (int s = 0; s < WSTEPS; ++s) {
      const int k = wave + 8 * s;
      if (k < WINS) __builtin_amdgcn_global_load_lds(wsrc + k * 64, (lds_ptr)(smem + WGT0 + buf * W_BYTES + k * 1024), 16, 0, 0);
    }
  };

  // ---- this lane's patches ----
  int g, i;
  patch_of(lane & 31, g, i);
  const bool has0 = wave < a.nslots, has1 = TSLOTS > 8 && wave + 8 < a.nslots;   // wave-uniform (tiles of <= 8 patches: one per wavefront)
  const int prow0 = 2 * wave + g, prow1 = 2 * (wave + 8) + g;
  // B operand addresses (K half 0; K half 1 = the same ^ 32): pixel q = (prow + ky) * 18 + i + kx, part = lane >> 5
  int bad0[9], bad1[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int q0 = (prow0 + t / 3) * CL_HW + i + t % 3, q1 = (prow1 + t / 3) * CL_HW + i + t % 3;
    bad0[t] = q0 * 64 + (((lane >> 5) ^ ((q0 >> 2) & 3)) << 4);
    bad1[t] = q1 * 64 + (((lane >> 5) ^ ((q1 >> 2) & 3)) << 4);
  }

  f32x16 acc[2][MT];

  // One chunk = 18 steps (tap t = s >> 1, K half kh = s & 1).  The fragments of step s + 2 are requested BEFORE the MFMAs of step
  // s and pinned there (hipcc otherwise puts every ds_read_b128 right in front of its MFMA behind an lgkmcnt(0): one exposed LDS
  // latency per MFMA -- the ISA of the first version); a ring of three register sets with compile-time slots.  A wavefront with
  // one patch alternates two accumulators (a single chain would serialise on the matrix pipe's result latency); they are added
  // in the epilogue.
  auto compute = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    if (!has0 || DBM_ABL_BIT(a, 1)) return;
    const unsigned char* ab = smem + buf * ACT_BYTES;
    const unsigned char* wb = smem + WGT0 + buf * W_BYTES + lane * 16;
    bf16x8 fa[3][MT], fb0[3], fb1[3];
    if (has1) {   // (wave-uniform: two patches share every weight fragment)
      auto ld = [&](int st, int slot) {
        const int t = st >> 1, kh = st & 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[slot][m] = *reinterpret_cast<const bf16x8*>(wb + ((t * 2 + kh) * MT + m) * 1024);
        fb0[slot] = *reinterpret_cast<const bf16x8*>(ab + (bad0[t] ^ (kh << 5)));
        fb1[slot] = *reinterpret_cast<const bf16x8*>(ab + (bad1[t] ^ (kh << 5)));
      };
      ld(0, 0);
      ld(1, 1);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        if (st + 2 < 18) ld(st + 2, (st + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st % 3][m], fb0[st % 3], acc[0][m], 0, 0, 0);
          acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st % 3][m], fb1[st % 3], acc[1][m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      auto ld = [&](int st, int slot) {
        const int t = st >> 1, kh = st & 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[slot][m] = *reinterpret_cast<const bf16x8*>(wb + ((t * 2 + kh) * MT + m) * 1024);
        fb0[slot] = *reinterpret_cast<const bf16x8*>(ab + (bad0[t] ^ (kh << 5)));
      };
      ld(0, 0);
      ld(1, 1);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        if (st + 2 < 18) ld(st + 2, (st + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[st & 1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st % 3][m], fb0[st % 3], acc[st & 1][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  // (One layer per launch.  The four 32-channel layers of a dense block as ONE launch whose tiles wait for their neighbours'
  //  previous layer through flag words -- `DBM_CL16_DENSE`, round 3 -- was bit for bit the same and not robustly faster (48-50 us
  //  against 59.6 for four launches on one box, 55 on another): removed in round 4.)
#ifndef CL16_NO_LAUNDER
  // (the lane constants are laundered through an empty asm: hipcc otherwise rematerialises the addresses of every chunk and tap)
#pragma unroll
  for (int s = 0; s < ASTEPS; ++s) asm volatile("" : "+v"(asrc[s]));
#pragma unroll
  for (int t = 0; t < 9; ++t) { asm volatile("" : "+v"(bad0[t])); asm volatile("" : "+v"(bad1[t])); }
#endif
  const int nchunk = a.Cin >> 5;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  stage(0, B0{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // chunk c is computed from buffer c & 1 while chunk c + 1 lands in the other one (two chunks per trip: the buffer
  // offsets are compile-time constants, so that hipcc can tell the DMA's destination from the fragment reads' source)
  for (int c = 0; c < (DBM_ABL_BIT(a, 4) ? 1 : nchunk); c += 2) {
    if (c + 1 < nchunk) stage(c + 1, B1{});
    compute(B0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c + 1 has landed (this wavefront's pieces; the barrier covers the rest)
    __syncthreads();
    if (c + 1 >= nchunk) break;
    if (c + 2 < nchunk) stage(c + 2, B0{});
    compute(B1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue ----
  // The accumulators hold (pixel = lane & 31, channels 4 (lane >> 5) + 8 (reg >> 2) + (reg & 3)): stored from there, a wave
  // instruction touches 32 pixels x 16 bytes -- thirty-two 16-byte pieces 384 / 256 bytes apart; measured (libdbm_measure.so) the
  // epilogue was 4.4 of the 14 us of a 64 -> 32 layer and 7+ of conv_layer5's.  So every patch goes through LDS once (the staging
  // buffers are free: all wavefronts have passed the last chunk's barrier; a wavefront only reads back what it wrote itself) and
  // leaves with the lanes of a pixel side by side: 8 MT lanes x 16 bytes = a pixel's whole 128 / 256-byte run per residual load
  // and fp32 store, its 64 / 128 bytes of bf16 per store, four or eight neighbouring pixels per instruction.
  if (DBM_ABL_BIT(a, 2)) return;
  if (has0 && !has1) {  // (one patch: even steps went to acc[0], odd steps to acc[1])
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][m][r] += acc[1][m][r];
  }
  constexpr int TROW = 32 * MT + 4;                 // floats per pixel row of the transpose tile (+ 16 bytes: conflict-free columns)
  constexpr int QPP = 8 * MT;                       // channel quads per pixel
  constexpr int NIT = 32 * QPP / 64;                // store iterations per patch (4 / 8)
  float* tbase = reinterpret_cast<float*>(smem);
  const int y0 = a.y0;
  // Round 5: vmcnt counts stores as well as loads, in order -- a load issued behind a store cannot be awaited without draining
  // that store (a write round trip).  The loop below used to load bias / r1 / r2 in EVERY iteration, behind the previous
  // iteration's stores, each load under its own (uniform) branch and hence awaited with vmcnt(0): eight drains + up to 24
  // serialised load round trips per wavefront (the "4.4 of 14 us" epilogue of round 4's ablation).  Now the bias quad is loaded
  // once (its channel depends on the lane only: QPP divides 64), the residual operands of a WHOLE patch are requested before the
  // patch's first store, and every access is a raw buffer access whose bounds check replaces the branches: offset -1 reads zero /
  // drops the store (pixels outside the plane), a zero-length buffer stands for an absent operand.
  auto rsrc = [](const void* ptr, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, ptr ? (int)bytes : 0, 0x00020000);   // (bytes < 2^31: launch_conv_cl16 checks)
  };
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const long npix = (long)a.N * a.H * a.W;
  const int co = 4 * (lane & (QPP - 1));            // this lane's channel quad, the same in every iteration
  const f4v bv = *reinterpret_cast<const f4v*>(a.bias + co);
  // (byte offsets relative to the IMAGE's first pixel: 32-bit for every plane the launcher accepts -- checked there)
  const __amdgpu_buffer_rsrc_t r1r = rsrc(a.r1 ? a.r1 + img * 64 : nullptr, 256L * a.H * a.W);
  const __amdgpu_buffer_rsrc_t r2r = rsrc(a.r2 ? a.r2 + img * 64 : nullptr, 256L * a.H * a.W);
  const __amdgpu_buffer_rsrc_t y32r = rsrc(a.y32 ? a.y32 + img * 64 : nullptr, 256L * a.H * a.W);
  const __amdgpu_buffer_rsrc_t y16r = rsrc(a.y16 ? a.y16 + img * a.yc : nullptr, 2L * a.yc * a.H * a.W);
  (void)npix;
  const bool resid = a.r1 != nullptr || a.r2 != nullptr;   // (uniform: conv_layer5)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (!(s == 0 ? has0 : has1)) continue;          // (wave-uniform)
    float* T = tbase + (size_t)(s == 0 ? wave : wave + 8) * 32 * TROW;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        *reinterpret_cast<f4v*>(T + (lane & 31) * TROW + 32 * m + 8 * rg + 4 * (lane >> 5)) =
            (f4v){acc[s][m][4 * rg], acc[s][m][4 * rg + 1], acc[s][m][4 * rg + 2], acc[s][m][4 * rg + 3]};
    const int prow_base = 2 * (s == 0 ? wave : wave + 8);
    int pixo[NIT];                                  // pixel index inside the image, -1: outside the plane
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int h = (it * 64 + lane) / QPP;
      int pg, pi;
      patch_of(h, pg, pi);
      const int gy = ty * TH + prow_base + pg, gx = tx * CL_TW + pi;
      pixo[it] = (gy < a.H && gx < a.W) ? gy * a.W + gx : -1;
    }
    f4v q1[NIT], q2[NIT];
    if (resid) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int off = pixo[it] >= 0 ? (pixo[it] * 64 + co) * 4 : -1;
        q1[it] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r1r, off, 0, 0));
        q2[it] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r2r, off, 0, 0));
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int h = (it * 64 + lane) / QPP;
      f4v v = *reinterpret_cast<const f4v*>(T + h * TROW + co);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bv[e];
      if (resid) {
        if (a.r1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = a.s1 * v[e] + q1[it][e];
        }
        if (a.r2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = a.s2 * v[e] + q2[it][e];
        }
      }
      if (a.act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : a.slope * v[e];
      }
      const int p = pixo[it];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), y32r, p >= 0 ? (p * 64 + co) * 4 : -1, 0, 0);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, o), y16r, p >= 0 ? (p * a.yc + y0 + co) * 2 : -1, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16 x 3") form for the layers ON the signal path of the sweep -- the two upsampling convolutions and the two
// offset convolutions of the deformable layers (srgan_train.py:488-514, 556-572): plain bf16 operands cost them 65-120 m rms
// at the reference's data range (DESIGN.md "bf16 at the data range"), fp32 MFMA runs at 1/16 of the bf16 rate.  Every fp32
// operand is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (sixteen significand bits), and a product is three bf16
// MFMAs: hi*hi + hi*lo + lo*hi (the dropped lo*lo term is 2^-18 relative) -- fp32 accumulation; 3/16 of the fp32 MFMA time at
// 2^-16 instead of 2^-9 operand precision.
//   * input: NHWC fp32 (64 channels), optionally the nearest x2 resize folded into the staging (source pixel = (y >> 1, x >> 1)
//     of the half-size plane); the split happens while a chunk of 16 channels is staged through registers (one 16-byte load
//     = 4 channels -> 8 bytes of hi and 8 bytes of lo, two ds_write_b64);
//   * LDS pixel record = 64 bytes [hi k0 | hi k1 | lo k0 | lo k1] with the same slot swizzle as above (conflict-free
//     ds_read_b128 fragment reads), weights [chunk][tap][mtile][hi | lo][lane][8] by LDS-DMA;
//   * output: NHWC fp32 (what the next layer of this kind and the fused deformable sampler read) or channel planes (the
//     offset tensors the deformable kernels consume).
// ---------------------------------------------------------------------------------------------------------------------
struct ClX3Args {
  const float* x; int xc;            // input NHWC fp32, xc channels per pixel; channels [0, Cin) are read
  int Cin;                           // multiple of 16
  int ups;                           // 1: x is the (H / 2, W / 2) plane, nearest x2 folded in
  const bf16x8* w;                   // packed [chunk][tap][mtile][hi | lo][lane][8]  (launch_pack_cl16x3)
  const float* bias;
  float* y32; int yc;                // NHWC fp32 output, yc channels per pixel (may be null)
  const float* r1; int r1c;          // residual added before the activation: NHWC fp32, r1c channels per pixel (may be null)
  __bf16* y16; int y16c;             // the output once more as bf16, NHWC with y16c channels per pixel (may be null)
  float* yp; long ysn; int ypc;      // channel-plane output yp[n * ysn + co * H * W + pixel], co < ypc (may be null)
  int act; float slope;
  int N, H, W, nslots, tilesX, tilesY;
#ifdef DBM_MEASURE
  int abl;                           // libdbm_measure.so only (DBM_CL16X3_ABL, results wrong): 1 no MFMA loop, 2 no epilogue, 4 no staging after chunk 0
#endif
};

__device__ __forceinline__ void split_bf16(const f4v v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
}

// TSLOTS: the largest tile height (in 2-row patches) the LDS layout is sized for.  12: one workgroup per CU.  8 (one output-channel
// tile only: 78 KB, <= 128 registers): TWO workgroups per CU -- the full-resolution planes are 13 rounds of tiles, and a tile's fixed
// part (index arithmetic, first chunk through registers, the barrier: 7 of its 16 us by DBM_CL16X3_ABL=7) then runs under the other
// workgroup's MFMAs.
template <int MT, int TSLOTS = CL_MAXSLOTS>
__global__ __launch_bounds__(CL_NT, TSLOTS <= 8 ? 2 : 1) void conv_cl16x3_kernel(ClX3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int TH = 2 * a.nslots, HR = TH + 2, NPIX = HR * CL_HW;
  constexpr int ACT_BYTES = cl_act_bytes(TSLOTS);
  constexpr int WINS = 9 * MT * 2;               // LDS-DMA instructions (1 KB) of a 16-channel chunk's weights
  constexpr int W_BYTES = WINS * 1024;
  constexpr int WGT0 = 2 * ACT_BYTES;

  int b = blockIdx.x;
  const int tx = b % a.tilesX; b /= a.tilesX;
  const int ty = b % a.tilesY;
  const int n = b / a.tilesY;
  const int gy0 = ty * TH - 1, gx0 = tx * CL_TW - 1;
  const int Hs = a.H >> a.ups, Ws = a.W >> a.ups;   // source plane
  const long simg = (long)n * Hs * Ws;

  // ---- activations: 16-byte pieces (4 channels) through registers; piece idx = (pixel q, quarter p) ----
  constexpr int ASTEPS = ((2 * TSLOTS + 2) * CL_HW * 4 + CL_NT - 1) / CL_NT;
  constexpr int WSTEPS = (WINS + 7) / 8;
  const float* asrc[ASTEPS];
  int ahi[ASTEPS];   // LDS byte of the piece's hi half (-1: no piece); lo half = the same ^ 32
#pragma unroll
  for (int s = 0; s < ASTEPS; ++s) {
    const int idx = tid + CL_NT * s;
    const int q = idx >> 2, p = idx & 3;
    const int hy = q / CL_HW, hx = q - hy * CL_HW;
    const int gy = gy0 + hy, gx = gx0 + hx;
    const bool have = q < NPIX;
    const bool inside = have && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    asrc[s] = inside ? a.x + (simg + (long)(gy >> a.ups) * Ws + (gx >> a.ups)) * a.xc + 4 * p : nullptr;
    ahi[s] = have ? q * 64 + ((((p >> 1)) ^ ((q >> 2) & 3)) << 4) + (p & 1) * 8 : -1;
  }
  f4v areg[ASTEPS];
  const int nchunk = a.Cin >> 4;
  auto issue = [&](int c, auto BUF) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < ASTEPS; ++s) areg[s] = asrc[s] ? *reinterpret_cast<const f4v*>(asrc[s] + 16 * c) : (f4v){0.f, 0.f, 0.f, 0.f};
    const u4v* wsrc = reinterpret_cast<const u4v*>(a.w) + (long)c * (WINS * 64) + lane;
#pragma unroll
    for (int s = 0; s < WSTEPS; ++s) {
      const int k = wave + 8 * s;
      if (k < WINS) __builtin_amdgcn_global_load_lds(wsrc + k * 64, (lds_ptr)(smem + WGT0 + buf * W_BYTES + k * 1024), 16, 0, 0);
    }
  };
  auto commit = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < ASTEPS; ++s) {
      if (ahi[s] < 0) continue;
      bf16x4 hi, lo;
      split_bf16(areg[s], hi, lo);
      *reinterpret_cast<bf16x4*>(smem + buf * ACT_BYTES + ahi[s]) = hi;
      *reinterpret_cast<bf16x4*>(smem + buf * ACT_BYTES + (ahi[s] ^ 32)) = lo;
    }
  };

  int g, i;
  patch_of(lane & 31, g, i);
  const bool has0 = wave < a.nslots, has1 = TSLOTS > 8 && wave + 8 < a.nslots;   // (tiles of <= 8 patches: one per wavefront)
  const int prow0 = 2 * wave + g, prow1 = 2 * (wave + 8) + g;
  int bad0[9], bad1[9];   // hi fragment of tap t (K half = lane >> 5); lo fragment = the same ^ 32
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int q0 = (prow0 + t / 3) * CL_HW + i + t % 3, q1 = (prow1 + t / 3) * CL_HW + i + t % 3;
    bad0[t] = q0 * 64 + (((lane >> 5) ^ ((q0 >> 2) & 3)) << 4);
    bad1[t] = q1 * 64 + (((lane >> 5) ^ ((q1 >> 2) & 3)) << 4);
  }
  f32x16 acc[2][MT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  auto compute = [&](auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    if (!has0 || DBM_ABL_BIT(a, 1)) return;
    const unsigned char* ab = smem + buf * ACT_BYTES;
    const unsigned char* wb = smem + WGT0 + buf * W_BYTES + lane * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      bf16x8 ah[MT], al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ah[m] = *reinterpret_cast<const bf16x8*>(wb + ((t * MT + m) * 2 + 0) * 1024);
        al[m] = *reinterpret_cast<const bf16x8*>(wb + ((t * MT + m) * 2 + 1) * 1024);
      }
      const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(ab + bad0[t]);
      const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(ab + (bad0[t] ^ 32));
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh0, acc[0][m], 0, 0, 0);   // (small terms first)
        acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl0, acc[0][m], 0, 0, 0);
        acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh0, acc[0][m], 0, 0, 0);
      }
      if (has1) {
        const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(ab + bad1[t]);
        const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(ab + (bad1[t] ^ 32));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh1, acc[1][m], 0, 0, 0);
          acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl1, acc[1][m], 0, 0, 0);
          acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh1, acc[1][m], 0, 0, 0);
        }
      }
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;

  issue(0, B0{});
  commit(B0{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < (DBM_ABL_BIT(a, 4) ? 1 : nchunk); c += 2) {
    if (c + 1 < nchunk) issue(c + 1, B1{});
    compute(B0{});
    if (c + 1 < nchunk) commit(B1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 >= nchunk) break;
    if (c + 2 < nchunk) issue(c + 2, B0{});
    compute(B1{});
    if (c + 2 < nchunk) commit(B0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue ----
  // (round 5: the bias quads are loaded ONCE, before the first store -- a load behind a store is awaited by draining the store,
  //  vmcnt being in order over both: the per-(patch, tile, quad) loads of the first version cost a write round trip each)
  if (DBM_ABL_BIT(a, 2)) return;
  const long plane = (long)a.H * a.W;
  f4v bq[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int e = 0; e < 4; ++e) bq[m][rg][e] = a.bias[32 * m + 8 * rg + 4 * (lane >> 5) + e];   // (dword loads: a bias tensor behind an
                                                                                                   //  18-float one is only 8-byte aligned)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (!(s == 0 ? has0 : has1)) continue;
    const int prow = s == 0 ? prow0 : prow1;
    const int gy = ty * TH + prow, gx = tx * CL_TW + i;
    if (gy >= a.H || gx >= a.W) continue;
    const long pin = (long)gy * a.W + gx;
    const long pix = (long)n * plane + pin;
    f4v rq[MT][4];   // the patch's residual quads: all requested before its first store
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        rq[m][rg] = a.r1 ? *reinterpret_cast<const f4v*>(a.r1 + pix * a.r1c + 32 * m + 8 * rg + 4 * (lane >> 5)) : (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int co = 32 * m + 8 * rg + 4 * (lane >> 5);
        f4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[s][m][4 * rg + e] + bq[m][rg][e] + rq[m][rg][e];
          if (a.act) v[e] = v[e] >= 0.f ? v[e] : a.slope * v[e];
        }
        if (a.y32) *reinterpret_cast<f4v*>(a.y32 + pix * a.yc + co) = v;
        if (a.y16) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
          *reinterpret_cast<bf16x4*>(a.y16 + pix * a.y16c + co) = o;
        }
        if (a.yp) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < a.ypc) a.yp[(long)n * a.ysn + (long)(co + e) * plane + pin] = v[e];
        }
      }
    }
  }
}

// dst[chunk][tap][mtile][hi | lo][lane][8]: W[cout = 32 mtile + (lane & 31)][cin = 16 chunk + 8 (lane >> 5) + e][tap] split in two
__global__ __launch_bounds__(256) void pack_cl16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int O, int C, int MT,
                                                          long total) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int i8 = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long r = e >> 9;
    const int part = (int)(r & 1); r >>= 1;
    const int m = (int)(r % MT); r /= MT;
    const int t = (int)(r % 9);
    const int chunk = (int)(r / 9);
    const int co = 32 * m + (lane & 31), ci = 16 * chunk + 8 * (lane >> 5) + i8;
    const float v = (co < O && ci < C) ? w[((long)co * C + ci) * 9 + t] : 0.f;
    const __bf16 h = (__bf16)v;
    dst[e] = part == 0 ? h : (__bf16)(v - (float)h);
  }
}

// dst[chunk][tap][k half][mtile][lane][8] = bf16(W[cout = 32 mtile + (lane & 31)][cin = 32 chunk + 16 khalf + 8 (lane >> 5) + e][tap])
__global__ __launch_bounds__(256) void pack_cl16_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int O, int C, int MT,
                                                        long total) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int i8 = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long r = e >> 9;
    const int m = (int)(r % MT); r /= MT;
    const int kh = (int)(r & 1); r >>= 1;
    const int t = (int)(r % 9);
    const int chunk = (int)(r / 9);
    const int co = 32 * m + (lane & 31), ci = 32 * chunk + 16 * kh + 8 * (lane >> 5) + i8;
    dst[e] = (__bf16)((co < O && ci < C) ? w[((long)co * C + ci) * 9 + t] : 0.f);
  }
}

// x (N, 64, plane) fp32 with image stride xsn  ->  res (N * plane, 64) fp32  and  act (N * plane, ac) bf16 channels 0..63
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float* __restrict__ x, long xsn, float* __restrict__ res,
                                                         __bf16* __restrict__ act, int ac, long total, int plane, int nch) {
  __shared__ float tile[64 * 65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long P0 = (long)blockIdx.x * 64;
  for (int c = wave; c < nch; c += 4) {   // coalesced along positions
    const long P = P0 + lane;
    float v = 0.f;
    if (P < total) {
      const long n = P / plane;
      v = x[n * xsn + (long)c * plane + (P - n * plane)];
    }
    tile[c * 65 + lane] = v;
  }
  __syncthreads();
  for (int p = wave; p < 64; p += 4) {    // coalesced along channels
    const long P = P0 + p;
    if (P >= total) break;
    if (lane >= nch) continue;
    const float v = tile[lane * 65 + p];
    if (res) res[P * 64 + lane] = v;
    if (act) act[P * ac + lane] = (__bf16)v;
  }
}

// res (N * plane, 64) fp32 -> y (N, 64, plane) fp32 with image stride ysn
__global__ __launch_bounds__(256) void cl_to_nchw_kernel(const float* __restrict__ res, float* __restrict__ y, long ysn, long total,
                                                         int plane, int nch) {
  __shared__ float tile[64 * 65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long P0 = (long)blockIdx.x * 64;
  for (int p = wave; p < 64; p += 4) {
    const long P = P0 + p;
    tile[lane * 65 + p] = P < total ? res[P * 64 + lane] : 0.f;
  }
  __syncthreads();
  for (int c = wave; c < nch; c += 4) {
    const long P = P0 + lane;
    if (P < total) {
      const long n = P / plane;
      y[n * ysn + (long)c * plane + (P - n * plane)] = tile[c * 65 + lane];
    }
  }
}

}  // namespace

size_t cl16_packed_elems(int Cin, int Cout) { return (size_t)(Cin / 32) * 9 * 2 * ((Cout + 31) / 32) * 64 * 8; }

void launch_pack_cl16(const float* w, void* dst, int O, int C, hipStream_t s) {
  DBM_CHECK(C % 32 == 0 && O >= 1 && O <= 64, "cl16 pack: Cin % 32 == 0, Cout <= 64");
  const int MT = (O + 31) / 32;
  const long total = (long)cl16_packed_elems(C, O);
  long nb = (total + 2047) / 2048;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(pack_cl16_kernel, dim3((unsigned)nb), dim3(256), 0, s, w, (__bf16*)dst, O, C, MT, total);
  DBM_HIP(hipGetLastError());
}

void launch_nchw_to_cl(const float* x, long xsn, float* res, void* act, int ac, int N, int plane, hipStream_t s, int nch) {
  const long total = (long)N * plane;
  DBM_CHECK(nch >= 1 && nch <= 64, "nchw_to_cl: 1..64 channels per call");
  hipLaunchKernelGGL(nchw_to_cl_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, x, xsn, res, (__bf16*)act, ac, total, plane,
                     nch);
  DBM_HIP(hipGetLastError());
}

void launch_cl_to_nchw(const float* res, float* y, long ysn, int N, int plane, hipStream_t s, int nch) {
  const long total = (long)N * plane;
  DBM_CHECK(nch >= 1 && nch <= 64, "cl_to_nchw: 1..64 channels per call");
  hipLaunchKernelGGL(cl_to_nchw_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, res, y, ysn, total, plane, nch);
  DBM_HIP(hipGetLastError());
}

// patches (of 2 rows) per tile: the plane in as few rounds of <= n_cus workgroups as possible, then the least work per SIMD
static int cl16_choose_slots(int N, int H, int W, int n_cus, int maxs = CL_MAXSLOTS, int per_cu = 1) {
  static const int forced = DBM_TUNE_GETENV("CL16_SLOTS") ? atoi(DBM_TUNE_GETENV("CL16_SLOTS")) : 0;
  if (forced >= 1 && forced <= maxs) return forced;
  const int tilesX = (W + CL_TW - 1) / CL_TW;
  int best = 8;
  double best_cost = 1e30;
  for (int ns = 2; ns <= maxs; ++ns) {
    const int tilesY = (H + 2 * ns - 1) / (2 * ns);
    const long wgs = (long)N * tilesX * tilesY;
    const long rounds = (wgs + (long)n_cus * per_cu - 1) / ((long)n_cus * per_cu);
    int load = 0;  // patches on the busiest SIMD: waves s and s + 4 share one, wave w owns patches w and w + 8
    for (int sd = 0; sd < 4; ++sd) {
      int l = 0;
      for (int p = sd; p < ns; p += 4) ++l;
      load = l > load ? l : load;
    }
    const double cost = (double)rounds * (1.0 + load);  // (one unit of fixed cost per tile: staging, barriers, epilogue)
    if (cost < best_cost - 1e-9) { best_cost = cost; best = ns; }
  }
  return best;
}

void launch_conv_cl16(const ClConvLaunch& L, hipStream_t s) {
  DBM_CHECK(L.Cin % 32 == 0 && L.Cin >= 32 && (L.Cout == 32 || L.Cout == 64), "cl16 conv: Cin % 32 == 0, Cout 32 or 64");
  DBM_CHECK(L.xc % 8 == 0 && (!L.y16 || (L.yc % 4 == 0 && L.y0 % 4 == 0)), "cl16 conv: channel strides must keep 16- / 8-byte alignment");
  // the epilogue addresses r1 / r2 / y32 / y16 through raw buffer accesses with 32-bit byte offsets relative to the image's first
  // pixel (and a 31-bit buffer size): refuse planes those cannot reach instead of dropping their stores
  DBM_CHECK(256L * L.H * L.W < (1L << 31) && (!L.y16 || 2L * L.yc * L.H * L.W < (1L << 31)),
            "cl16 conv: one image plane must stay below 2 GiB per operand (32-bit epilogue offsets)");
  static const int n_cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
  }();
  ClConvArgs a;
  a.x = (const __bf16*)L.x; a.xc = L.xc; a.Cin = L.Cin; a.w = (const bf16x8*)L.w; a.bias = L.bias;
  a.y16 = (__bf16*)L.y16; a.yc = L.yc; a.y0 = L.y0; a.y32 = L.y32;
  a.r1 = L.r1; a.s1 = L.s1; a.r2 = L.r2; a.s2 = L.s2; a.act = L.act; a.slope = L.slope;
  a.N = L.N; a.H = L.H; a.W = L.W;
  const int MT = L.Cout / 32;
  a.tilesX = (L.W + CL_TW - 1) / CL_TW;
  // (Tiles of at most eight patches with two workgroups per CU were measured in round 3, on that round's kernel, and bought nothing.)
  // (round 6, again: tiles of <= 8 patches with one output-channel tile -- 78 KB, two workgroups per CU -- for launches of several
  //  rounds of tiles: crops batched per forward.  DBM_CL16_PAIR, libdbm_measure.so: 0 never)
  static const int pair_env = DBM_TUNE_GETENV("CL16_PAIR") ? atoi(DBM_TUNE_GETENV("CL16_PAIR")) : 1;
  // (DBM_CL16_PAIR_MIN: smallest number of 16-row tiles that takes the form.  A single crop -- 324 such tiles -- is slower with it:
  //  4.51 against 4.47 ms; conv_layer5's 64 output channels as two workgroups of 32 per tile lost as well: continent 1.669 against 1.683 s
  //  without any pairing, where the 32-channel layers alone gain 3 %: round6_calls/40_conv_cl16_halves.patch)
  static const long pair_min = DBM_TUNE_GETENV("CL16_PAIR_MIN") ? atol(DBM_TUNE_GETENV("CL16_PAIR_MIN")) : 4L * n_cus;
  const bool pair = pair_env && MT == 1 && (long)L.N * a.tilesX * ((L.H + 15) / 16) >= pair_min;
  a.nslots = pair ? cl16_choose_slots(L.N, L.H, L.W, n_cus, 8, 2) : cl16_choose_slots(L.N, L.H, L.W, n_cus);
  a.tilesY = (L.H + 2 * a.nslots - 1) / (2 * a.nslots);
  size_t lds = 2 * (size_t)cl_act_bytes(pair ? 8 : CL_MAXSLOTS) + 2 * (size_t)18 * MT * 1024;
  a.zeros = L.zeros;
#ifdef DBM_MEASURE
  static const int abl = DBM_MEASURE_ENV("CL16_ABL");
  a.abl = abl;
#endif
  DBM_CHECK(L.zeros != nullptr, "cl16 conv: a device zero block is required");
  static bool attr = false;
  if (!attr) {
    DBM_HIP(hipFuncSetAttribute((const void*)conv_cl16_kernel<1, CL_MAXSLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)conv_cl16_kernel<2, CL_MAXSLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)conv_cl16_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr = true;
  }
  const unsigned grid = (unsigned)((long)L.N * a.tilesX * a.tilesY);
  if (g_profiler.enabled) {
    // algorithmic bytes: Cin bf16 channels per pixel in, the packed weights, the outputs (bf16 and / or the 64-channel fp32
    // residual stream) and the fp32 residual operands
    const double px = (double)L.N * L.H * L.W;
    const double bytes = px * (2.0 * L.Cin + (L.y16 ? 2.0 * L.Cout : 0.0) + (L.y32 ? 256.0 : 0.0) + (L.r1 ? 256.0 : 0.0) + (L.r2 ? 256.0 : 0.0)) +
                         2.0 * 9 * L.Cin * L.Cout;
    char tag[40];
    snprintf(tag, sizeof(tag), "cl16_c%d>%d_%dx%d_n%d", L.Cin, L.Cout, L.H, L.W, L.N);
    g_profiler.begin(s, 0, 2.0 * px * L.Cout * L.Cin * 9, bytes, tag, grid);
  }
  if (pair)
    hipLaunchKernelGGL((conv_cl16_kernel<1, 8>), dim3(grid), dim3(CL_NT), lds, s, a);
  else if (MT == 1)
    hipLaunchKernelGGL((conv_cl16_kernel<1, CL_MAXSLOTS>), dim3(grid), dim3(CL_NT), lds, s, a);
  else
    hipLaunchKernelGGL((conv_cl16_kernel<2, CL_MAXSLOTS>), dim3(grid), dim3(CL_NT), lds, s, a);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

size_t cl16x3_packed_elems(int Cin, int Cout) { return (size_t)(Cin / 16) * 9 * ((Cout + 31) / 32) * 2 * 64 * 8; }

void launch_pack_cl16x3(const float* w, void* dst, int O, int C, hipStream_t s) {
  DBM_CHECK(C % 16 == 0 && O >= 1 && O <= 64, "cl16x3 pack: Cin % 16 == 0, Cout <= 64");
  const int MT = (O + 31) / 32;
  const long total = (long)cl16x3_packed_elems(C, O);
  long nb = (total + 2047) / 2048;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(pack_cl16x3_kernel, dim3((unsigned)nb), dim3(256), 0, s, w, (__bf16*)dst, O, C, MT, total);
  DBM_HIP(hipGetLastError());
}

void launch_conv_cl16x3(const ClX3Launch& L, hipStream_t s) {
  DBM_CHECK(L.Cin % 16 == 0 && L.Cin >= 16 && L.Cout >= 1 && L.Cout <= 64, "cl16x3 conv: Cin % 16 == 0, Cout <= 64");
  DBM_CHECK(L.xc % 4 == 0 && (!L.y32 || L.yc % 4 == 0), "cl16x3 conv: channel strides must keep 16-byte alignment");
  DBM_CHECK(!L.ups || (L.H % 2 == 0 && L.W % 2 == 0), "cl16x3 conv: the folded resize doubles both sides");
  static const int n_cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
  }();
  ClX3Args a;
  a.x = L.x; a.xc = L.xc; a.Cin = L.Cin; a.ups = L.ups; a.w = (const bf16x8*)L.w; a.bias = L.bias;
  a.y32 = L.y32; a.yc = L.yc; a.r1 = L.r1; a.r1c = L.r1c; a.y16 = (__bf16*)L.y16; a.y16c = L.y16c; a.yp = L.yp; a.ysn = L.ysn; a.ypc = L.ypc; a.act = L.act; a.slope = L.slope;
  DBM_CHECK(!L.r1 || (L.r1c % 4 == 0 && L.Cout % 32 == 0), "cl16x3 conv: the residual needs whole 32-channel tiles and 16-byte alignment");
  a.N = L.N; a.H = L.H; a.W = L.W;
#ifdef DBM_MEASURE
  a.abl = DBM_MEASURE_ENV("CL16X3_ABL");
#endif
  const int MT = (L.Cout + 31) / 32;
  a.tilesX = (L.W + CL_TW - 1) / CL_TW;
  // two workgroups per CU (tiles of <= 8 patches, one output-channel tile) once the plane is several rounds of tiles anyway
  static const int pair_env = DBM_TUNE_GETENV("CL16X3_PAIR") ? atoi(DBM_TUNE_GETENV("CL16X3_PAIR")) : 1;   // (0: one workgroup per CU -- A/B)
  const bool pair = pair_env && MT == 1 && (long)L.N * a.tilesX * ((L.H + 15) / 16) >= 4L * n_cus;
  a.nslots = pair ? cl16_choose_slots(L.N, L.H, L.W, n_cus, 8, 2) : cl16_choose_slots(L.N, L.H, L.W, n_cus);
  a.tilesY = (L.H + 2 * a.nslots - 1) / (2 * a.nslots);
  const size_t lds = 2 * (size_t)cl_act_bytes(pair ? 8 : CL_MAXSLOTS) + 2 * (size_t)9 * MT * 2 * 1024;
  static bool attr = false;
  if (!attr) {
    DBM_HIP(hipFuncSetAttribute((const void*)conv_cl16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)conv_cl16x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBM_HIP(hipFuncSetAttribute((const void*)conv_cl16x3_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr = true;
  }
  const unsigned grid = (unsigned)((long)L.N * a.tilesX * a.tilesY);
  if (g_profiler.enabled) {
    // algorithmic bytes: the fp32 NHWC source plane (a quarter of the output plane when the resize is folded in), hi + lo
    // weight images, Cout fp32 channels out.  flops: the layer's, not the three MFMAs the split spends on each
    const double px = (double)L.N * L.H * L.W;
    const double bytes = px * (4.0 * L.Cin / (L.ups ? 4 : 1) + 4.0 * L.Cout) + 2.0 * 2.0 * 9 * L.Cin * L.Cout;
    char tag[40];
    snprintf(tag, sizeof(tag), "x3_c%d>%d_%dx%d_n%d%s", L.Cin, L.Cout, L.H, L.W, L.N, L.ups ? "u" : "");
    g_profiler.begin(s, 0, 2.0 * px * L.Cout * L.Cin * 9, bytes, tag, grid);
  }
  if (pair)
    hipLaunchKernelGGL((conv_cl16x3_kernel<1, 8>), dim3(grid), dim3(CL_NT), lds, s, a);
  else if (MT == 1)
    hipLaunchKernelGGL(conv_cl16x3_kernel<1>, dim3(grid), dim3(CL_NT), lds, s, a);
  else
    hipLaunchKernelGGL(conv_cl16x3_kernel<2>, dim3(grid), dim3(CL_NT), lds, s, a);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}
