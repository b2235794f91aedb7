// Internal declarations shared by the HIP translation units of libdbm.so (gfx950 only).
#pragma once
// Measurement switches that SKIP work (results are then wrong) exist only in libdbm_measure.so (make MEASURE=1, built by
// tools/build_measure.sh with -DDBM_MEASURE): in the product library the queries below are the constant 0, the branches behind
// them are compiled out and the switch names do not occur in the binary.
//   dbm_abl_skip(): bit mask of kernel classes that are NOT launched (1 BatchNorm, 4 Adam, 8 pair folds, 32 linear layers,
//                   128 few-channel convs, 256 im2col, 512 per-layer implicit GEMMs): what a class costs INSIDE the step
//   DBM_MEASURE_ENV(name): atoi of an environment switch (weight gradients off, chain / cl16 kernel ablations, no repack)
#include <cstdlib>
#ifdef DBM_MEASURE
inline int dbm_measure_env(const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; }
#define DBM_MEASURE_ENV(name) dbm_measure_env("DBM_" name)
#define DBM_ABL_BIT(a, m) ((a).abl & (m))
#else
#define DBM_MEASURE_ENV(name) 0
#define DBM_ABL_BIT(a, m) false
#endif
inline int dbm_abl_skip() { static const int v = DBM_MEASURE_ENV("ABL_SKIP"); return v; }
// TUNING switches (launch-size rules, kernel-form overrides, schedule variants whose A/Bs read "the default stands": profiles/r5/
// ab_igemm_knobs_late.txt, ab_wgrad_knobs_late.txt, tune_igemm.txt): they select valid kernels, but nothing in the product or its tests
// sets them -- round 6 moved them out of libdbm.so.  DBM_TUNE_GETENV("X") is getenv("DBM_X") in libdbm_measure.so (tools/tune_igemm.py,
// tools/experiments/ab_env.sh with DBM_LIB) and a null pointer in the product library, where the defaults are compile-time constants and
// the names do not occur (tests/test_abi.py lists the names the product library does read and fails on any nobody exercises).
#ifdef DBM_MEASURE
#define DBM_TUNE_GETENV(name) getenv("DBM_" name)
#else
inline const char* dbm_no_env() { return nullptr; }
#define DBM_TUNE_GETENV(name) dbm_no_env()
#endif

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <stdexcept>

#define DBM_MAX_TAPS 16

struct DbmError : std::runtime_error {
  int code;
  DbmError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define DBM_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      throw DbmError(2, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + \
                            ":" + std::to_string(__LINE__));                              \
  } while (0)

#define DBM_CHECK(cond, msg)                                                   \
  do {                                                                         \
    if (!(cond)) throw DbmError(1, std::string(msg) + " (" #cond ") @" + __FILE__ + ":" + \
                                       std::to_string(__LINE__));              \
  } while (0)

// ----------------------------------------------------------------------------------------------
// Implicit-GEMM convolution descriptor (igemm.hip).  One launch computes, for every logical output
// position (n, a, b), a in [0,OHl), b in [0,OWl), and every output channel c < Cout:
//   acc = sum_t sum_ci  wp[t][ci][c] * x[n][ci][(a*sin + dy[t]) >> ups][(b*sin + dx[t]) >> ups]
// (terms whose logical input coordinate falls outside [0, Hin<<ups) x [0, Win<<ups) are zero), then
//   v = s1*(acc + bias[c]) + (c < r1_nch ? r1s*r1 : 0);  if (r2) v = s2*v + r2;  if (accumulate) v += y;
//   if (act) v = lrelu(v);  if (mask && c >= mask_c0) v *= lrelu'(mask)
// and stores v at y[n][c][(a*so+oy0)*OWp + (b*so+ox0)].
// r1, r2, mask are addressed like y (channel stride ysc, same spatial index).
// ----------------------------------------------------------------------------------------------
struct ConvDesc {
  const float* x;
  long xsn;       // elements between images of x
  int xsc;        // elements between channels of x (Hin*Win)
  int Cin;        // multiple of 32
  int cin_live;   // 0, or the number of leading input channels that can be non-zero (a multiple of 8 <= Cin): the data gradient of a layer
                  // whose output channels were padded (the 18 -> 32 offset tensors) -- conv_tile.hip skips the chunks behind it
  int Hin, Win;   // physical input dims
  int ups;        // 0/1: nearest x2 upsample folded into the gather
  int N, OHl, OWl;
  int sin;
  int T;
  signed char dy[DBM_MAX_TAPS];
  signed char dx[DBM_MAX_TAPS];
  signed char tmap[DBM_MAX_TAPS];  // set by the launcher (conv_tile.hip): window cell (ky, kx), row-major -> tap index t
  int abl;        // libdbm_measure.so only (conv_tile.hip): 1 = only chunk 0 is staged, 2 = no MFMAs, 4 = no epilogue (results wrong)
  const float* wp;  // packed [T][Cin][CoutP]
  int CoutP;        // multiple of 32
  int Cout;
  const float* bias;
  float* y;
  long ysn;
  int ysc;
  int OWp, so, oy0, ox0;
  float s1;
  const float* r1;
  long r1sn;
  int r1_nch;
  float r1s;
  const float* r2;
  long r2sn;
  float s2;
  int accumulate;
  int act;
  float slope;
  const float* mask;
  long masksn;
  int mask_c0;
  float* yt;      // optional second copy of the output, channels-last (N * OHl * OWl, 64): the fused deformable sampler's input layout.
                  // Written by conv_tile.hip only (Cout == 64, plain epilogue): ask conv_tile_writes_yt(d) first.
  const float* ch_scale;  // per-output-channel multiplier applied BEFORE the bias (null: 1): v = acc * ch_scale[c] + bias[c] -- an
                          // eval-mode BatchNorm folded into the convolution (ch_scale = gamma / sqrt(avg_var + eps), bias = beta -
                          // avg_mean * ch_scale; discriminator.hip)
  const float* zeros;  // >= 4 bytes of device zeros
  unsigned planeM, owM; // set by the launcher: floor(2^32 / (OHl*OWl)), floor(2^32 / OWl) (division-free position decode)
  const void* wp16;    // bf16 forward image [T][Cin/16][2][CoutP][8] (null: fp32 MFMA path)
  int ksplit;          // set by the launcher: > 1 = blockIdx.z % ksplit owns Cin / ksplit input channels (few-tile, long-K
                       // layers: the deep discriminator convs).  Deterministic: every workgroup writes its partial tile to
                       // ks_part, the LAST one to arrive at the tile's counter sums the slices in index order and runs the
                       // epilogue.  (DBM_IGEMM_KSPLIT=2: the older form, fp32 atomics onto a pre-zeroed y, plain layers only.)
  float* ks_part;      // set by the launcher: [tile][ksplit][32 x 32] partial tiles (workspace of the launch stream)
  unsigned* ks_cnt;    // set by the launcher: one arrival counter per tile, zero between launches
  // Merged phases of a stride-2 data gradient (T == 4): blockIdx.z / ksplit = phase ph = 2 py + px, whose four taps are
  // dy/dx[4 ph ..], whose weight image is phwp[ph], output offset (py, px) and logical plane phOH[ph] x phOW[ph]
  // (the planes differ by one row / column when the gradient's dims are odd).  nphase <= 1: a single-phase launch.
  int nosplit;         // set by the launcher: every wavefront owns a position tile over the whole K (blockIdx.x counts groups
                       // of WAVES tiles), no cross-wavefront reduction
  int pm_groups;       // set by the launcher: > 0 = position-major tiles (igemm_pm_kernel): 32-image groups per output position
  int pm_gshift;       // ... log2(sets of four channel pairs per live tap and wavefront)
  int nphase;
  const float* phwp[4];
  short phOH[4], phOW[4];
  unsigned phPlaneM[4], phOwM[4];
};

void launch_igemm_conv(const ConvDesc& d, hipStream_t s);
// the LDS-tiled form for the mid-size training planes (conv_tile.hip); launch_igemm_conv dispatches to it
int conv_tile_plan(ConvDesc& d, long* wgs);
void conv_tile_launch(const ConvDesc& d, int cfg, hipStream_t s);
bool conv_tile_writes_yt(const ConvDesc& d);   // will launch_igemm_conv(d) fill d.yt?

// Per-kernel-family timing with HIP events on the launch stream (bench.py's roofline leg).
// family 0 = igemm_conv_kernel (forward + data gradient), 1 = weight-gradient kernels, 2 = trunk_fused_kernel,
// 3 = trunk_fused_bwd_kernel, 4 = trunk_fused_kernel with a helper workgroup per image (counted as 2 when four are asked for).
struct KernelProfiler {
  bool enabled = false;
  bool serial = false;   // dbm_profile_begin_serial: the host synchronises the device around every bracketed launch
  // bytes: ALGORITHMIC bytes of the launch (operands read once + results written once: what a perfect cache hierarchy would
  // move); tag: a short label of the launch's shape (layer geometry) for the per-shape table of bench.py / profiles
  struct Rec { hipEvent_t a, b; double flops, bytes; int family; long wgs; char tag[40]; };
  std::vector<Rec> recs;
  void begin(hipStream_t s, int family, double flops, double bytes = 0.0, const char* tag = nullptr, long wgs = 0);  // wgs: workgroups of the (first) launch -- joins a bracket to a rocprofv3 row
  std::string dump_records();  // "family flops bytes ms wgs tag\n" per bracket, then clears (dbm_profile_records)
  void end(hipStream_t s);
  void collect(double* out, int nfam);  // [ms, flops, launches] for families 0 .. nfam-1, then clears
  // phase marks: one event per named point of a training step on the main stream (dbm_phase_marks)
  bool marks_enabled = false;
  std::vector<std::pair<std::string, hipEvent_t>> marks;
  void mark(hipStream_t s, const char* name);
  std::string dump_marks();  // "name ms_since_first_mark\n" per mark, then clears
};
#define DBM_MARK(s, name) do { if (g_profiler.marks_enabled) g_profiler.mark((s), (name)); } while (0)
extern KernelProfiler g_profiler;

// weight packing (igemm.hip): dst[t][k][mP] with (k,m) = (cin,cout) (transpose=0) or (cout,cin) (transpose=1);
// tap t reads OIHW element (ky[t], kx[t]).  Rows m >= M are zero-filled; k in [K, KP) zero-filled.
// All layers of a model are packed by ONE launch driven by a device-resident job table.
struct PackJob {
  const float* w;
  float* dst;
  int O, C, KH, KW, T, transpose, KP, MP;
  signed char ky[DBM_MAX_TAPS], kx[DBM_MAX_TAPS];
  int block_start, block_count;
};
void launch_pack_jobs(const PackJob* d_jobs, int njobs, int total_blocks, hipStream_t s);

// wgrad (wgrad.hip): gW[o][c][ky][kx] (+)= scale * sum_{n,a,b} dy[n][o][a][b] * x[n][c][(a*stride+ky-pad)>>ups][...]
struct WgradDesc {
  const float* x;   // input activations
  long xsn; int xsc; int Cin; int Hin, Win; int ups;
  const float* dy;  // output gradients
  long dysn; int dysc; int Cout; int OH, OW;
  int KH, KW, stride, pad;
  int N;
  float scale;
  float* gW;        // canonical OIHW, accumulated with atomicAdd
  float* gb;        // may be null; accumulated with atomicAdd
};
void launch_wgrad(const WgradDesc& d, hipStream_t s);  // single layer, synchronous (tests / op-level API)

struct WgradPlan {
  WgradDesc d;
  int groups, coutTiles, S;  // workgroups = groups (input-channel groups) x coutTiles x S (K splits)
  int wg_count;
  int G;        // 32-channel input tiles per workgroup (= active wavefronts)
  int IB, R;    // band = IB images x R output rows
  int nbr;      // row-bands per image
  int BP, BPp;  // positions per band, padded to even
  int YS;       // LDS row stride of the dy slab (odd)
  int Rin, Wst; // staged logical input patch rows / cols per image
  int ImgS;     // Rin*Wst
  int XS;       // LDS channel stride of the patch (odd)
  int nbands;
  // whole-image bands of plain (not upsampled) planes: both operands are contiguous per image and are staged with
  // 16-byte loads; `fast` = 0 falls back to the per-element gather
  int fast;
  unsigned planeM, winM;  // ceil(2^32 / (Hin*Win)), ceil(2^32 / Win): exact quotients for the sizes staged here
  unsigned oplaneM;       // ceil(2^32 / (OH*OW))
  int wave_task;          // 0: wgrad_kernel (workgroup form), 2: wgrad_wave_dma_kernel, 3: wgrad_band_dma_kernel,
                          // 4: wgrad_direct_kernel (Wst = segments per output row, nbands = segments, S = K slices)
  const float* zeros;     // >= 4 bytes of device zeros (out-of-image rows of the row-band DMA form)
  // deterministic folding (no fp32 atomics): partial[slice][cout tile][group][wave][tap][16][64] in accumulator order,
  // partial_b[slice][cout tile][32]; summed in slice order by wgrad_fold_kernel.  null = atomics.
  float* partial;
  float* partial_b;
  int fold_start;         // first workgroup of this layer in the fold launch
  // wavefront slot w of a workgroup owns input tile grp * fold_ctmul + (w % fold_cts) (if (w % fold_cts) < fold_ctmul) and
  // the fold_tpw taps from (w / fold_cts) * fold_tpw
  int fold_slots, fold_cts, fold_ctmul, fold_tpw;
  // Pair buffers (the other deterministic folding): K slices 2k and 2k + 1 add with atomics into buffer k, which is
  // zero and has gW's own layout (+ Cout bias floats): two contributions commute bit for bit.  pair_direct: pair 0 goes
  // straight to gW / gb (the gradient is known to be zero).  wgrad_pair_fold_kernel adds the buffers in order, clears them.
  float* pairW;
  long pair_stride;   // floats per buffer: Cout * Cin * T, then Cout
  int pair_n, pair_direct;
};
extern bool g_wgrad_deterministic;  // dbm_set_deterministic: weight gradients are folded without fp32 atomics
// fills p, returns the dynamic LDS bytes it needs (0: not eligible for the wave-task form)
size_t wgrad_plan(const WgradDesc& d, WgradPlan& p, int level = 0, int wave_task = 0, int S_fixed = 0);

// All weight gradients of one backward pass: collected as descriptors, planned and uploaded once per
// workspace shape, then launched as one kernel per kernel size (1x1, 3x3, 4x4).
struct WgradBatch {
  std::vector<WgradDesc> descs;
  bool built = false;
  bool built_deterministic = false;
  // The caller guarantees that every gW / gb of this batch is ZERO when the batch runs and receives nothing else (the
  // training step: cleargrads, then one backward pass).  Two K slices may then fold with atomics even in deterministic
  // mode: (0 + a) + b == (0 + b) + a bit for bit -- no partial tiles, no fold kernel for those layers.
  bool cleared_target = false, built_cleared = false;
  static const int NCAT = 10;  // see WgradBatch::build
  WgradPlan* d_plans[NCAT] = {};
  int* d_starts[NCAT] = {};
  int nplans[NCAT] = {}, total_wg[NCAT] = {};
  size_t lds[NCAT] = {};
  float* d_partial[NCAT] = {};   // scratch of the atomic-free folding (per category)
  int fold_wgs[NCAT] = {};
  bool pair_mode[NCAT] = {};     // deterministic folding through pair buffers (WgradPlan::pairW)
  double flops[NCAT] = {};
  double abytes[NCAT] = {};      // algorithmic bytes of a category's launch: every layer's x, dy and gW once
  // 4x4 stride-2 layers on tiny planes (wgrad_s2tiny_kernel): their own plan table, outside the categories
  void* d_tiny = nullptr;
  int n_tiny = 0, tiny_wgs = 0, tiny_rowlen = 4;
  double tiny_flops = 0.0, tiny_bytes = 0.0;
  std::vector<int> tiny_owner;
  void add(const WgradDesc& d) { if (!built) descs.push_back(d); }
  void build();
  void launch(hipStream_t s);
  void reset();
  ~WgradBatch() { reset(); }
};
