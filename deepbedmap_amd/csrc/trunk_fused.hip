// The whole RRDB trunk of the generator forward (srgan_train.py:333-360 ResidualDenseBlock.forward, :393-404
// ResInResDenseBlock.forward, :546 residual_network) on 9x9 planes as ONE persistent launch.
//
// Layer by layer the 9x9 stage is a chain of 5 * 3 * n_rrdb launches of 2-12 us of matrix work each; a kernel boundary
// plus prologue / epilogue costs more than the MFMAs between two of them.  Here a workgroup owns three rows of one image
// (27 positions = one 32-wide MFMA column tile) for the whole trunk:
//  * the dense block's concat (192 channels x (3 own + 2 halo rows)) lives in LDS, zero-framed, so the B operand of every
//    tap is one ds_read_b32 with an immediate offset and no border logic;
//  * the K axis (input channels) is split over eight wavefronts in units of four channels; channel quad q belongs to
//    wavefront q % 8 in every layer, so a wavefront is the only reader of "its" planes and fetches their halo rows alone;
//  * weights stream from L2 in the order each wavefront consumes them (pack_trunk_fused_kernel), 16 bytes per lane per
//    load, one unit (4 channels x 9 taps = 18 MFMAs) ahead, across layer and dense-block boundaries;
//  * the three workgroups of an image exchange the boundary rows of every layer's output through data-tagged 8-byte
//    granules ({value, layer tag}, one relaxed agent-scope store / load each: no fences, no flags).  A layer's K loop runs
//    the old channels and the newest channels' middle-row taps first; only the last 12 MFMAs per wavefront need the halo;
//  * split-K partial tiles are reduced through LDS; the epilogue (bias, LeakyReLU, `a5*rs + a0`, `a3*rs + x`) writes the
//    LDS planes, the global concat buffers the backward pass reads (training) and the neighbours' granules.
// Every spin is bounded (an error word is raised instead of a hang); workgroups never wait for anything but the two
// neighbours of their own image.  241 VGPRs: one workgroup per CU, a 64-image launch (192 workgroups) is resident at once on
// the 256 CUs; kernels of other streams only delay it (they finish).
#include "model.h"

namespace {

constexpr int CS = 56;           // floats per LDS channel plane: 5 rows x (9 + 1 shared zero column) + frame / dummy lanes
constexpr int NWAVE = 8;
constexpr int NTHREADS = NWAVE * 64;
constexpr int UNIT = 1152;       // floats of one (unit, out-channel tile) weight block: 64 lanes x 18
constexpr int WAVE_RDB = 26 * UNIT;  // (2+3+4+5) + 2*6 blocks per wavefront per dense block
constexpr int SPIN_LIMIT = 1 << 21;

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

extern __shared__ float lds[];   // act[192][CS] | red[8][16][64]
constexpr int RED0 = 192 * CS;
constexpr size_t LDS_BYTES = (size_t)(RED0 + NWAVE * 16 * 64) * sizeof(float);

struct Args {
  const float* wstream;   // [wave][rdb][26][UNIT]: every wavefront reads one contiguous stream
  const float* bstream;   // [rdb][192]: conv1..4 (32 each), conv5 (64)
  const float* in;        // (N, 192, 81) concat buffer of dense block 0; channels 0..63 hold the trunk input
  float* cat[TRUNK_FUSED_MAXCAT];  // training: nrdb + 1 concat buffers (cat[0] == in), in the kernel arguments: a
                          // pointer table in global memory costs a dependent vector load per epilogue
  int store_all;
  float* out;             // inference: concat buffer receiving the trunk output in channels 0..63
  unsigned long long* inbox;  // [3 * images][2][2][64][9] granules {value, tag}
  int* err;
  int* err_dev;
  int nrdb, nimg, img0, epoch;
  float rs, slope;
};

struct Wave {
  int lane, w, t;
  int img, band, cl;       // image (global index), row band 0..2, cluster index inside this launch
  int bofs;                // lane's B base: (lane >> 5) * CS + position offset
  int n;                   // position 0..31 (>= 27: padding lane)
  int pofs;                // 11 + position offset (own row cell of a plane), valid for n < 27
  const float* wp;         // next weight block of this wavefront
  float xres[4];           // RRDB input at this thread's four conv5 outputs
  // epilogue constants of this thread's first output (register r = 2 w, tile 0); the others add immediates
  unsigned ep_l;           // LDS cell of (channel m0, own position)
  unsigned ep_g;           // element offset inside a concat buffer: (img * 192 + m0) * 81 + band * 27 + n
  unsigned ep_up, ep_dn;   // granule index inside the upper / lower neighbour's inbox (parity 0, channel m0)
  unsigned hl_g[2];        // halo fetch: granule index of slot s = lane (+64) inside my inbox (parity 0, channel 0)
  int hl_l[2];             // ... and its LDS cell relative to the quad's first plane (-1: nothing to fetch)
  bool st_ok, up_ok, dn_ok;
};

#define DI __device__ __forceinline__

template <int NM> DI void issue_loads(float (&A)[36], const float* p, int lane) {
#pragma unroll
  for (int mt = 0; mt < NM; ++mt) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f4v v = *reinterpret_cast<const f4v*>(p + mt * UNIT + c * 256 + lane * 4);
      A[mt * 18 + 4 * c + 0] = v.x; A[mt * 18 + 4 * c + 1] = v.y; A[mt * 18 + 4 * c + 2] = v.z; A[mt * 18 + 4 * c + 3] = v.w;
    }
    const f2v u = *reinterpret_cast<const f2v*>(p + mt * UNIT + 1024 + lane * 2);
    A[mt * 18 + 16] = u.x; A[mt * 18 + 17] = u.y;
  }
}

// SEL 0: all nine taps; 1: the middle kernel row (needs no halo row); 2: the outer kernel rows
template <int NM, int SEL> DI void mma_taps(const float (&A)[36], int b, f16v (&acc)[2]) {
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (SEL == 1 && tap / 3 != 1) continue;
    if (SEL == 2 && tap / 3 == 1) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float bv = lds[b + ks * 2 * CS + (tap / 3) * 10 + tap % 3];
#pragma unroll
      for (int mt = 0; mt < NM; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[mt * 18 + tap * 2 + ks], bv, acc[mt], 0, 0, 0);
    }
  }
}

DI unsigned long long granule_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DI void granule_store(unsigned long long* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// One layer of a dense block.  K = 0..4 (conv_layer1..5).  A0 / A1: the weight ping-pong (static parity per dense block).
template <int K>
__device__ __forceinline__ void dense_layer(const Args& a, Wave& W, float (&A0)[36], float (&A1)[36], int j, bool last_rdb) {
  constexpr int NM = K == 4 ? 2 : 1;
  constexpr int U = 2 + K;
  constexpr int base = K == 0 ? 0 : K == 1 ? 2 : K == 2 ? 5 : K == 3 ? 9 : 14;
  const int lane = W.lane, w = W.w;
  const int serial = j * 5 + K;                      // layer serial inside the trunk
  const unsigned tag_in = ((unsigned)a.epoch << 12) | (unsigned)serial;  // what the producer (layer serial - 1) wrote
  const int par_in = (serial - 1) & 1;
  const bool need_halo = serial > 0;                 // the first layer's halo rows came with the input load

  // hipcc's waitcnt bookkeeping does not survive branches (epilogue stores, the rare spin path): wherever it is unsure it
  // emits vmcnt(0).  So every unit first takes the wait for ITS weights (issued one unit ago), and only then issues the
  // next unit's loads: a conservative vmcnt(0) never sits behind a freshly issued prefetch.
  {
    float (&first)[36] = (base & 1) ? A1 : A0;
#pragma unroll
    for (int i = 0; i < 18 * NM; ++i) asm volatile("" ::"v"(first[i]));
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- layer-start requests, pinned here: the halo granules of this wavefront's newest quads (checked just before
  // use) and the biases of this thread's epilogue outputs.  Every lane loads (lanes without a granule re-read slot 0).
  // granule slot s (0..71) of a quad: channel e = s / 18, side = (s / 9) & 1 (0: row above, 1: row below), column s % 9
  constexpr int NQ = K == 0 ? 2 : 1;                 // newest quads of this wavefront
  const unsigned long long* gp[NQ][2];
  unsigned long long gv[NQ][2];
  int gdst[NQ][2];
  {
    const unsigned long long* inb = a.inbox + (par_in ? 2 * 64 * 9 : 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int quad = K == 0 ? w + 8 * q : w + 8 * (1 + K);
      const int pch = K == 0 ? 4 * (w + 8 * q) : 4 * w;  // channel of the producing layer's output
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        gdst[q][r] = (need_halo && W.hl_l[r] >= 0) ? W.hl_l[r] + quad * 4 * CS : -1;
        gp[q][r] = inb + (W.hl_g[r] + pch * 9);
        gv[q][r] = granule_load(gp[q][r]);
      }
    }
  }
  float bias[2 * NM];
#pragma unroll
  for (int k = 0; k < 2 * NM; ++k) {
    const int r = 2 * w + (k & 1);
    bias[k] = a.bstream[j * 192 + (K < 4 ? 32 * K : 128) + (k >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
  }
  __builtin_amdgcn_sched_barrier(0);

  f16v acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool par = ((base + u) & 1) != 0;
    float (&cur)[36] = par ? A1 : A0;
    float (&nxt)[36] = par ? A0 : A1;
    // the next unit of this wavefront's stream (next layer / next dense block included; the stream is padded at its end)
    const bool next_is_nm2 = (u + 1 < U) ? (K == 4) : (K == 3);
    if (u > 0) {
#pragma unroll
      for (int i = 0; i < 18 * NM; ++i) asm volatile("" ::"v"(cur[i]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (next_is_nm2) { issue_loads<2>(nxt, W.wp, lane); W.wp += 2 * UNIT; }
    else { issue_loads<1>(nxt, W.wp, lane); W.wp += UNIT; }
    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the prefetch to just before its first use
    const int b = W.bofs + (w + 8 * u) * 4 * CS;
    const bool newest = (K == 0) || (u == U - 1);
    if (!newest) {
      mma_taps<NM, 0>(cur, b, acc);
    } else {
      mma_taps<NM, 1>(cur, b, acc);
      // pin the middle-row MFMAs BEFORE the wait (hipcc otherwise sinks them below the spin loop: they are pure)
      asm volatile("" : "+v"(acc[0]));
      if (NM == 2) asm volatile("" : "+v"(acc[1]));
      const int q = K == 0 ? u : 0;
      bool ok = true;
#pragma unroll
      for (int r = 0; r < 2; ++r) ok = ok && (gdst[q][r] < 0 || (unsigned)(gv[q][r] >> 32) == tag_in);
      if (!__all(ok)) {  // rare: the neighbour is behind
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (gdst[q][r] >= 0) {
            int spins = 0;
#pragma nounroll
            while ((unsigned)(gv[q][r] >> 32) != tag_in) {
              __builtin_amdgcn_s_sleep(1);
              gv[q][r] = granule_load(gp[q][r]);
              if (++spins > SPIN_LIMIT) { *a.err = 1; *a.err_dev = 1; break; }
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (gdst[q][r] >= 0) lds[gdst[q][r]] = __uint_as_float((unsigned)gv[q][r]);
      mma_taps<NM, 2>(cur, b, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- split-K reduction over the eight wavefronts + epilogue, one 32-channel tile at a time ----
  const unsigned tag_out = ((unsigned)a.epoch << 12) | (unsigned)(serial + 1);
  const int par_out = serial & 1;
  // uniform bases of this layer's outputs
  float* gbase = nullptr;
  if (K < 4) { if (a.store_all) gbase = a.cat[j] + (64 + 32 * K) * 81; }
  else { gbase = a.store_all ? a.cat[j + 1] : (last_rdb ? a.out : nullptr); }
  unsigned long long* obox = a.inbox + (par_out ? 2 * 64 * 9 : 0);
  const bool publish = !(K == 4 && last_rdb);
  constexpr int LCH = K < 4 ? (64 + 32 * K) * CS : 0;  // first LDS plane of this layer's output
#pragma unroll
  for (int mt = 0; mt < NM; ++mt) {
    if (mt) __syncthreads();  // the previous tile's sums have been read
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[RED0 + (w * 16 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    float v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 2 * w + i;
      float t = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWAVE; ++ww) t += lds[RED0 + (ww * 16 + r) * 64 + lane];
      t += bias[mt * 2 + i];
      if (K < 4) {
        t = t >= 0.f ? t : a.slope * t;
      } else {
        t = a.rs * t + lds[W.ep_l + (mt * 32 + i) * CS];  // a5 * rs + a0  (:358)
        if (j % 3 == 2) {                                 // a3 * rs + x   (:402)
          t = a.rs * t + W.xres[mt * 2 + i];
          W.xres[mt * 2 + i] = t;
        }
      }
      v[i] = t;
    }
    if (W.st_ok) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lds[LCH + W.ep_l + (mt * 32 + i) * CS] = v[i];
        if (gbase) gbase[W.ep_g + (mt * 32 + i) * 81] = v[i];
      }
      if (publish) {
        if (W.up_ok) {  // my top row is the bottom halo of the band above
#pragma unroll
          for (int i = 0; i < 2; ++i) granule_store(obox + (W.ep_up + (mt * 32 + i) * 9), v[i], tag_out);
        }
        if (W.dn_ok) {
#pragma unroll
          for (int i = 0; i < 2; ++i) granule_store(obox + (W.ep_dn + (mt * 32 + i) * 9), v[i], tag_out);
        }
      }
    }
  }
  __syncthreads();  // planes written: the next layer may read them
}

__global__ __launch_bounds__(512) void trunk_fused_kernel(Args a) {
  Wave W;
  W.t = threadIdx.x; W.lane = W.t & 63; W.w = W.t >> 6;
  // block b runs on XCD b % 8 (observed; speed only): the three bands of an image are blocks b, b + 8, b + 16
  const int B = blockIdx.x;
  W.cl = (B / 24) * 8 + (B % 8);
  W.band = (B / 8) % 3;
  if (W.cl >= a.nimg) return;
  W.img = a.img0 + W.cl;
  W.n = W.lane & 31;
  {
    const int n = W.n;
    const int po = n < 27 ? (n / 9) * 10 + n % 9 : (n == 27 ? 9 : n == 28 ? 19 : 29 + (n - 29));
    W.bofs = (W.lane >> 5) * CS + po;
    W.pofs = 11 + po;
  }
  W.wp = a.wstream + (size_t)W.w * a.nrdb * WAVE_RDB;
  {
    const int m0 = ((2 * W.w) & 3) + 8 * ((2 * W.w) >> 2) + 4 * (W.lane >> 5), n = W.n;
    const unsigned me = (unsigned)(W.cl * 3 + W.band);
    W.st_ok = n < 27;
    W.up_ok = n < 9 && W.band > 0;
    W.dn_ok = n >= 18 && n < 27 && W.band < 2;
    W.ep_l = (unsigned)(m0 * CS + W.pofs);
    W.ep_g = (unsigned)((W.img * 192 + m0) * 81 + W.band * 27 + n);
    W.ep_up = (((me - 1) * 2) * 2 + 1) * 576 + m0 * 9 + n;        // [wg][parity][side][64][9]
    W.ep_dn = (((me + 1) * 2) * 2 + 0) * 576 + m0 * 9 + (n - 18);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int s0 = W.lane + 64 * r;
      const int sl = s0 < 72 ? s0 : 0;
      const int e = sl / 18, side = (sl / 9) & 1, c = sl % 9;
      const bool have = s0 < 72 && (side == 0 ? W.band > 0 : W.band < 2);
      W.hl_g[r] = ((me * 2) * 2 + side) * 576 + e * 9 + c;
      W.hl_l[r] = have ? e * CS + (side ? 40 : 0) + c + 1 : -1;
    }
  }

  for (int i = W.t; i < 192 * CS; i += NTHREADS) lds[i] = 0.f;
  __syncthreads();
  // trunk input: channels 0..63, rows 3*band-1 .. 3*band+3 of this image
  const float* in = a.in + (size_t)W.img * 192 * 81;
  for (int i = W.t; i < 64 * 45; i += NTHREADS) {
    const int ch = i / 45, rc = i % 45, r = rc / 9, c = rc % 9;
    const int row = 3 * W.band - 1 + r;
    if (row >= 0 && row < 9) lds[ch * CS + r * 10 + c + 1] = in[ch * 81 + row * 9 + c];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // this thread's conv5 outputs: tile mt = k / 2, register r = 2 w + (k & 1)
    const int r = 2 * W.w + (k & 1);
    const int m = (k >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (W.lane >> 5);
    W.xres[k] = W.n < 27 ? in[m * 81 + W.band * 27 + W.n] : 0.f;
  }
  __syncthreads();

  float A0[36], A1[36];
  issue_loads<1>(A0, W.wp, W.lane);
  W.wp += UNIT;
  for (int j = 0; j < a.nrdb; ++j) {
    const bool last = j == a.nrdb - 1;
    dense_layer<0>(a, W, A0, A1, j, last);
    dense_layer<1>(a, W, A0, A1, j, last);
    dense_layer<2>(a, W, A0, A1, j, last);
    dense_layer<3>(a, W, A0, A1, j, last);
    dense_layer<4>(a, W, A0, A1, j, last);
  }
}

// dst-driven gather of the trunk's weights into the per-wavefront streams (one launch per optimizer step)
__global__ void pack_trunk_fused_kernel(const float* const* wsrc, const float* const* bsrc, float* wstream, float* bstream,
                                        int nrdb) {
  const long total = (long)nrdb * NWAVE * WAVE_RDB;
  for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (long)gridDim.x * blockDim.x) {
    const int w = (int)(f / ((long)nrdb * WAVE_RDB));
    const int j = (int)((f / WAVE_RDB) % nrdb);
    const int x0 = (int)(f % WAVE_RDB);
    const int bidx = x0 / UNIT, x = x0 % UNIT;
    int K, u, mt = 0;
    if (bidx < 2) { K = 0; u = bidx; }
    else if (bidx < 5) { K = 1; u = bidx - 2; }
    else if (bidx < 9) { K = 2; u = bidx - 5; }
    else if (bidx < 14) { K = 3; u = bidx - 9; }
    else { K = 4; u = (bidx - 14) / 2; mt = (bidx - 14) % 2; }
    int lane, i;
    if (x < 1024) { lane = (x % 256) / 4; i = 4 * (x / 256) + x % 4; }
    else { lane = (x - 1024) / 2; i = 16 + (x - 1024) % 2; }
    const int tap = i / 2, ks = i % 2;
    const int cin = 4 * (w + 8 * u) + 2 * ks + (lane >> 5);
    const int cout = mt * 32 + (lane & 31);
    const int Cin = 64 + 32 * K;
    wstream[f] = wsrc[j * 5 + K][((long)cout * Cin + cin) * 9 + tap];
  }
  const long nb = (long)nrdb * 192;
  for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nb; f += (long)gridDim.x * blockDim.x) {
    const int j = (int)(f / 192), c = (int)(f % 192);
    const int K = c < 128 ? c / 32 : 4;
    bstream[f] = bsrc[j * 5 + K][c - 32 * K];
  }
}

// ------------------------------------------------------------------------------------------------------------------
size_t trunk_fused_stream_floats(int nrdb) { return (size_t)nrdb * NWAVE * WAVE_RDB + 4 * UNIT; }  // + read-ahead pad
size_t trunk_fused_inbox_bytes(int nimg) { return (size_t)3 * nimg * 2 * 2 * 64 * 9 * sizeof(unsigned long long); }

void launch_pack_trunk_fused(const float* const* d_wsrc, const float* const* d_bsrc, float* wstream, float* bstream, int nrdb,
                             hipStream_t s) {
  hipLaunchKernelGGL(pack_trunk_fused_kernel, dim3(1024), dim3(256), 0, s, d_wsrc, d_bsrc, wstream, bstream, nrdb);
  DBM_HIP(hipGetLastError());
}

void launch_trunk_fused(const TrunkFusedLaunch& L, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    DBM_HIP(hipFuncSetAttribute((const void*)trunk_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr = true;
  }
  Args a;
  a.wstream = L.wstream; a.bstream = L.bstream; a.in = L.in; a.out = L.out;
  a.store_all = L.cat != nullptr;
  for (int i = 0; i < TRUNK_FUSED_MAXCAT; ++i) a.cat[i] = (L.cat && i <= L.nrdb) ? L.cat[i] : nullptr;
  DBM_CHECK(L.nrdb + 1 <= TRUNK_FUSED_MAXCAT, "fused trunk: too many dense blocks");
  a.inbox = L.inbox; a.err = L.err; a.err_dev = L.err_dev;
  a.nrdb = L.nrdb; a.nimg = L.nimg; a.img0 = L.img0; a.epoch = L.epoch & 0xFFFFF; a.rs = L.rs; a.slope = L.slope;
  const int grid = ((L.nimg + 7) / 8) * 24;
  const double flop = 2.0 * 19408896.0 * L.nrdb * L.nimg;  // 19 408 896 MAC per dense block and tile
  if (g_profiler.enabled) g_profiler.begin(s, 2, flop);
  hipLaunchKernelGGL(trunk_fused_kernel, dim3(grid), dim3(NTHREADS), LDS_BYTES, s, a);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}
