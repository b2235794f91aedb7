// DeepbedmapInputBlock (srgan_train.py:223-266) on the training tile -- X 11x11, W1 110x110, W2 2 x 22x22, W3 11x11 -> four valid
// convolutions of 32 channels each on the 9x9 grid, concatenated -- as ONE launch.
//
// Layer by layer the block is eight dependent launches (two single-channel 3x3 convolutions, im2col + GEMM for the 30x30 / stride-10
// and the 6x6 / stride-2 branch) of 6-34 us each: 0.13-0.19 ms at the head of BOTH generator forwards of a training iteration, i.e. on
// its critical path twice, for 0.33 GFLOP.  Here a workgroup (eight wavefronts) owns three output rows of one image -- the trunk
// kernels' band: 27 of a tile's 32 MFMA columns -- stages the input rows those reach in LDS (W1: 50 contiguous rows of 110 = 22 KB) and
// forms every branch as v_mfma_f32_32x32x2_f32 tiles (rows = the branch's 32 output channels) whose B operand is one ds_read_b32 at
// `position base + immediate`: no im2col image.  The 900-long K axis of the W1 branch is split over six wavefronts (five kernel rows =
// 75 MFMAs each, partial tiles reduced through LDS in wavefront order: deterministic); the seventh forms the W2 branch (36 MFMAs), the
// eighth X and W3 (5 each).  A operands: the packed forward images of the two wide branches (IgLayer::wf, [k][32]: one coalesced dword
// per lane and MFMA) and the OIHW tensors of the 3x3 branches.
// The wide branches' weight gradients still read an im2col image: a retained pass rebuilds it off the critical path
// (Generator::backward, side stream).
#include <cstdint>
#include <cstdio>
#include "kernels.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int OW = 9, BAND = 3, TP = BAND * OW;   // 27 positions per workgroup
constexpr int W1W = 110, W1S = 10, W1K = 30;      // conv_on_W1: 30x30, stride 10
constexpr int W2W = 22, W2S = 2, W2K = 6;         // conv_on_W2: 6x6, stride 2, two input channels
constexpr int XW = 11;                            // conv_on_X / conv_on_W3: 3x3
constexpr int R1 = (BAND - 1) * W1S + W1K;        // 50 input rows of W1 per band
constexpr int R2 = (BAND - 1) * W2S + W2K;        // 10 of W2
constexpr int RX = BAND + 2;                      // 5 of X / W3
constexpr int NW1 = 6, KY_PER_WAVE = W1K / NW1;   // K split of the W1 branch: five kernel rows per wavefront
constexpr int STEPS1 = KY_PER_WAVE * W1K / 2;     // 75 MFMAs (two k per MFMA: the lane half picks the odd kernel column)

__global__ __launch_bounds__(512) void input_block_fused_kernel(const InputBlockLaunch a) {
  __shared__ __attribute__((aligned(16))) float in1[R1 * W1W];
  __shared__ __attribute__((aligned(16))) float in2[2 * R2 * W2W];
  __shared__ float inx[2][64];
  __shared__ float red[NW1][16][64];
  const int band = blockIdx.x, n = blockIdx.y, oy0 = BAND * band;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- stage: whole input rows, contiguous in memory; every request is out before the first LDS store ----
  {
    constexpr int Q1 = R1 * W1W / 4, Q2 = R2 * W2W / 4;   // 16-byte pieces: 1375 of W1, 55 per W2 channel
    const f4v* g1 = reinterpret_cast<const f4v*>(a.w1 + (long)n * (W1W * W1W) + oy0 * W1S * W1W);
    const f4v t0 = g1[tid], t1 = g1[tid + 512];
    const int i2 = tid + 1024 < Q1 ? tid + 1024 : Q1 - 1;
    const f4v t2 = g1[i2];
    const int c = tid < Q2 ? 0 : 1, i = tid < 2 * Q2 ? tid - c * Q2 : 0;
    const f4v u = reinterpret_cast<const f4v*>(a.w2 + ((long)n * 2 + c) * (W2W * W2W) + oy0 * W2S * W2W)[i];
    const int ix = (tid & 63) < RX * XW ? (tid & 63) : 0;
    const float vx = (wave == 2 ? a.x : a.w3)[(long)n * (XW * XW) + oy0 * XW + ix];
    reinterpret_cast<f4v*>(in1)[tid] = t0;
    reinterpret_cast<f4v*>(in1)[tid + 512] = t1;
    reinterpret_cast<f4v*>(in1)[i2] = t2;
    if (tid < 2 * Q2) reinterpret_cast<f4v*>(in2)[tid] = u;
    if (wave == 2 || wave == 3) inx[wave - 2][tid & 63] = vx;
  }
  __syncthreads();
  const int p = lane & 31, hh = lane >> 5;
  const int pv = p < TP ? p : 0;   // (padding columns repeat position 0; never stored)
  const int oyr = pv / OW, ox = pv - OW * oyr;
  float* yb = a.y + (long)n * a.ysn + TP * band + p;
  // rows of this lane's 16 accumulator registers: (r & 3) + 8 (r >> 2) + 4 hh
  auto store_tile = [&](const f16v& acc, int ch0, const float* bias) {
    float b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * hh];
    if (p < TP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) yb[(long)(ch0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * (OW * OW)] = acc[r] + b[r];
    }
  };
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (wave < NW1) {
    // k = (KY_PER_WAVE wave + kyr) * 30 + 2 j + hh  ->  packed row k, column o = lane & 31: element 32 k + o = const + 64 step + lane
    const float* A = a.wf1 + (long)wave * (KY_PER_WAVE * W1K) * 32 + lane;
    const float* B = in1 + (oyr * W1S + wave * KY_PER_WAVE) * W1W + ox * W1S + hh;
    float av[STEPS1];
#pragma unroll
    for (int s = 0; s < STEPS1; ++s) av[s] = A[64 * s];
#pragma unroll
    for (int kyr = 0; kyr < KY_PER_WAVE; ++kyr)
#pragma unroll
      for (int j = 0; j < W1K / 2; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kyr * (W1K / 2) + j], B[kyr * W1W + 2 * j], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  } else if (wave == NW1) {
    // conv_on_W2: k = c * 36 + ky * 6 + kx = 2 s + hh (kx even + hh: never crosses a kernel row)
    const float* A = a.wf2 + lane;
    const float* B = in2 + oyr * W2S * W2W + ox * W2S + hh;
    float av[W2K * W2K];
#pragma unroll
    for (int s = 0; s < W2K * W2K; ++s) av[s] = A[64 * s];
#pragma unroll
    for (int s = 0; s < W2K * W2K; ++s) {
      const int k = 2 * s, c = k / (W2K * W2K), ky = (k % (W2K * W2K)) / W2K, kx = k % W2K;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], B[c * (R2 * W2W) + ky * W2W + kx], acc, 0, 0, 0);
    }
    store_tile(acc, 64, a.b2);
  } else {
    // conv_on_X, then conv_on_W3: k = ky * 3 + kx = 2 s + hh, k = 9 is padding (A = 0)
#pragma unroll
    for (int br = 0; br < 2; ++br) {
      const float* wt = (br ? a.w3w : a.wx) + (lane & 31) * 9;
      const float* B = inx[br] + oyr * XW + ox;
      float av[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int k = 2 * s + hh;
        av[s] = k < 9 ? wt[k] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
        const int off = hh ? (k1 / 3) * XW + k1 % 3 : (k0 / 3) * XW + k0 % 3;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], B[off], acc, 0, 0, 0);
      }
      store_tile(acc, br ? 96 : 0, br ? a.b3 : a.bx);
    }
  }
  __syncthreads();
  // ---- conv_on_W1: the six partial tiles, summed in wavefront order ----
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + 512 * q, r = idx >> 6, l = idx & 63;
    float v = red[0][r][l];
#pragma unroll
    for (int w = 1; w < NW1; ++w) v += red[w][r][l];
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    if (col < TP) a.y[(long)n * a.ysn + (long)(32 + row) * (OW * OW) + TP * band + col] = v + a.b1[row];
  }
}


// ---- the same block on LARGE planes (the area sweep's crops: 288 x 288 -> 286 x 286): a workgroup owns 32 consecutive positions of ONE
// output row and stages the input window those reach (W1: 30 rows x 340 columns = 41 KB; W2: 2 x 6 x 68; X / W3: 3 x 34).  Replaces, per
// crop, an im2col image of 304 MB (written, then read by the GEMM), the 72-row one of the W2 branch and four launches.  Same wavefront
// roles and K order as the band kernel above. ----
constexpr int RW_P = 32;                              // positions per workgroup
constexpr int RW_L1 = (RW_P - 1) * W1S + W1K;         // 340 columns of W1
constexpr int RW_L2 = (RW_P - 1) * W2S + W2K;         // 68 of W2
constexpr int RW_LX = RW_P + 2 + 2;                   // 34 of X / W3, padded to 36

__global__ __launch_bounds__(512) void input_block_rows_kernel(const InputBlockLaunch a, int H, int W) {
  __shared__ __attribute__((aligned(16))) float in1[W1K * RW_L1];
  __shared__ float in2[2 * W2K * RW_L2];
  __shared__ float inx[2][3 * RW_LX];
  __shared__ float red[NW1][16][64];
  const int OWp = W - 2, OHp = H - 2;
  const int ox0 = blockIdx.x * RW_P, oy = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- stage the windows (columns past the plane's right edge: zero; their positions are never stored) ----
  {
    const int W10 = W1S * W, avail1 = W10 - W1S * ox0;   // floats left in an input row from the window's first column
    const float* g1 = a.w1 + (long)n * W10 * (W1S * H) + (long)(W1S * oy) * W10 + W1S * ox0;
    if ((W10 & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w1) & 15) == 0) {   // 16-byte pieces (W even): 85 per row
      constexpr int Q = RW_L1 / 4;
      f4v t[5];
      int di[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int i = tid + 512 * u;
        const bool in = i < W1K * Q;
        const int r = in ? i / Q : 0, c4 = in ? i - r * Q : 0;
        const bool ok = in && 4 * c4 + 3 < avail1;
        const f4v v = *reinterpret_cast<const f4v*>(g1 + (long)r * W10 + (ok ? 4 * c4 : 0));
        t[u] = ok ? v : (f4v){0.f, 0.f, 0.f, 0.f};
        di[u] = in ? i : -1;
      }
#pragma unroll
      for (int u = 0; u < 5; ++u)
        if (di[u] >= 0) reinterpret_cast<f4v*>(in1)[di[u]] = t[u];
    } else {
      for (int i0 = tid; i0 < W1K * RW_L1; i0 += 512 * 4) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 512 * u < W1K * RW_L1 ? i0 + 512 * u : i0;
          const int r = i / RW_L1, c = i - r * RW_L1;
          const float v = g1[(long)r * W10 + (c < avail1 ? c : 0)];
          t[u] = c < avail1 ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + 512 * u < W1K * RW_L1) in1[i0 + 512 * u] = t[u];
      }
    }
    const int W2w = W2S * W, avail2 = W2w - W2S * ox0;
    for (int i = tid; i < 2 * W2K * RW_L2; i += 512) {
      const int c = i / (W2K * RW_L2), rem = i - c * (W2K * RW_L2), r = rem / RW_L2, col = rem - r * RW_L2;
      const float v = a.w2[(((long)n * 2 + c) * (W2S * H) + W2S * oy + r) * W2w + W2S * ox0 + (col < avail2 ? col : 0)];
      in2[i] = col < avail2 ? v : 0.f;
    }
    const int availx = W - ox0;
    if (tid < 2 * 3 * RW_LX) {
      const int b = tid / (3 * RW_LX), rem = tid - b * (3 * RW_LX), r = rem / RW_LX, col = rem - r * RW_LX;
      const float v = (b ? a.w3 : a.x)[((long)n * H + oy + r) * W + ox0 + (col < availx ? col : 0)];
      inx[b][rem] = col < availx ? v : 0.f;
    }
  }
  __syncthreads();
  const int p = lane & 31, hh = lane >> 5;
  const bool pok = ox0 + p < OWp;
  const int pv = pok ? p : 0;
  const long plane = (long)OHp * OWp;
  float* yb = a.y + (long)n * a.ysn + (long)oy * OWp + ox0 + p;
  // channels-last output (a.yt): registers 4 g .. 4 g + 3 of a lane are four consecutive channels of its position -- one 16-byte store
  float* ytb = a.yt ? a.yt + (((long)n * OHp + oy) * OWp + ox0 + p) * 128 : nullptr;
  auto store_tile = [&](const f16v& acc, int ch0, const float* bias) {
    float b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * hh];
    if (pok) {
      if (ytb) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f4v*>(ytb + ch0 + 8 * g + 4 * hh) =
              (f4v){acc[4 * g] + b[4 * g], acc[4 * g + 1] + b[4 * g + 1], acc[4 * g + 2] + b[4 * g + 2], acc[4 * g + 3] + b[4 * g + 3]};
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) yb[(long)(ch0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * plane] = acc[r] + b[r];
      }
    }
  };
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (wave < NW1) {
    const float* A = a.wf1 + (long)wave * (KY_PER_WAVE * W1K) * 32 + lane;
    const float* B = in1 + (wave * KY_PER_WAVE) * RW_L1 + pv * W1S + hh;
    float av[STEPS1];
#pragma unroll
    for (int s = 0; s < STEPS1; ++s) av[s] = A[64 * s];
#pragma unroll
    for (int kyr = 0; kyr < KY_PER_WAVE; ++kyr)
#pragma unroll
      for (int j = 0; j < W1K / 2; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kyr * (W1K / 2) + j], B[kyr * RW_L1 + 2 * j], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  } else if (wave == NW1) {
    const float* A = a.wf2 + lane;
    const float* B = in2 + pv * W2S + hh;
    float av[W2K * W2K];
#pragma unroll
    for (int s = 0; s < W2K * W2K; ++s) av[s] = A[64 * s];
#pragma unroll
    for (int s = 0; s < W2K * W2K; ++s) {
      const int k = 2 * s, c = k / (W2K * W2K), ky = (k % (W2K * W2K)) / W2K, kx = k % W2K;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], B[c * (W2K * RW_L2) + ky * RW_L2 + kx], acc, 0, 0, 0);
    }
    store_tile(acc, 64, a.b2);
  } else {
#pragma unroll
    for (int br = 0; br < 2; ++br) {
      const float* wt = (br ? a.w3w : a.wx) + (lane & 31) * 9;
      const float* B = inx[br] + pv;
      float av[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int k = 2 * s + hh;
        av[s] = k < 9 ? wt[k] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
        const int off = hh ? (k1 / 3) * RW_LX + k1 % 3 : (k0 / 3) * RW_LX + k0 % 3;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], B[off], acc, 0, 0, 0);
      }
      store_tile(acc, br ? 96 : 0, br ? a.b3 : a.bx);
    }
  }
  __syncthreads();
  if (a.yt) {   // channels-last: a thread sums one register quad (four consecutive channels) of one lane's position
    if (tid < 256) {
      const int g = tid >> 6, l = tid & 63, col = l & 31;
      f4v v;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float t = red[0][4 * g + rr][l];
#pragma unroll
        for (int w = 1; w < NW1; ++w) t += red[w][4 * g + rr][l];
        v[rr] = t + a.b1[rr + 8 * g + 4 * (l >> 5)];
      }
      if (ox0 + col < OWp) *reinterpret_cast<f4v*>(a.yt + (((long)n * OHp + oy) * OWp + ox0 + col) * 128 + 32 + 8 * g + 4 * (l >> 5)) = v;
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + 512 * q, r = idx >> 6, l = idx & 63;
    float v = red[0][r][l];
#pragma unroll
    for (int w = 1; w < NW1; ++w) v += red[w][r][l];
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    if (ox0 + col < OWp) a.y[(long)n * a.ysn + (long)(32 + row) * plane + (long)oy * OWp + ox0 + col] = v + a.b1[row];
  }
}

}  // namespace

bool input_block_fused_ok(int H, int W) { return H == XW && W == XW; }
bool input_block_rows_ok(int H, int W) { return H >= 3 && W - 2 >= 2 * RW_P && (long)(H - 2) <= 65535; }   // (planes at least two tiles wide)

void launch_input_block_rows(const InputBlockLaunch& a, int H, int W, hipStream_t s) {
  if (a.N <= 0) return;
  const int OWp = W - 2, OHp = H - 2;
  const dim3 grid((OWp + RW_P - 1) / RW_P, OHp, a.N);
  if (g_profiler.enabled) {
    char tag[40];
    snprintf(tag, sizeof(tag), "input_block_%dx%d_n%d", H, W, a.N);
    const double px = (double)a.N * OHp * OWp;
    g_profiler.begin(s, 0, 2.0 * 32 * 990 * px, 4.0 * (a.N * (double)H * W * (1.0 + 100 + 8 + 1) + 128 * px + 32 * 990 + 128), tag,
                     (long)grid.x * grid.y * grid.z);
  }
  hipLaunchKernelGGL(input_block_rows_kernel, grid, dim3(512), 0, s, a, H, W);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}

void launch_input_block_fused(const InputBlockLaunch& a, hipStream_t s) {
  if (a.N <= 0) return;
  if (g_profiler.enabled) {   // four valid convolutions: 32 x (9 + 900 + 72 + 9) MACs per position; inputs + weights in, 128 channels out
    char tag[40];
    snprintf(tag, sizeof(tag), "input_block_11x11_n%d", a.N);
    g_profiler.begin(s, 0, 2.0 * 32 * 990 * 81.0 * a.N, 4.0 * (a.N * (121.0 + 12100 + 968 + 121 + 128 * 81) + 32 * 990 + 128), tag,
                     (OW / BAND) * a.N);
  }
  hipLaunchKernelGGL(input_block_fused_kernel, dim3(OW / BAND, (unsigned)a.N), dim3(512), 0, s, a);
  if (g_profiler.enabled) g_profiler.end(s);
  DBM_HIP(hipGetLastError());
}
