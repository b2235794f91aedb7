// Data-gradient chain of the RRDB trunk (backward of srgan_train.py:333-360 / :393-404) on 9x9 planes as persistent
// launches, one per group of residual-in-residual blocks (the weight-gradient batches of a group go to the side stream
// as soon as its launch is enqueued behind it).
//
// Same decomposition as trunk_fused.hip: a workgroup owns three rows of one image, the gradient of the dense block's
// concat (192 channels) lives in LDS, zero-framed; the three workgroups of an image exchange the boundary rows of every
// finished 32-channel gradient block through data-tagged granules.  Differences:
//  * a layer's data gradient has FEW input channels (K = 32 x 9, conv_layer5: 64 x 9) and MANY outputs (64..192): the
//    output is cut into 16-channel x 16-position sub-tiles (v_mfma_f32_16x16x4_f32), sub-tile i belongs to wavefront
//    i % 8 and is computed over the whole K by that wavefront alone -- no split-K, no reduction; the epilogue is a
//    read-modify-write of the wavefront's own LDS cells;
//  * partial sums (the accumulated gradient of channels that still receive contributions) never leave LDS; a 32-channel
//    block goes to global memory once, when it is final (masked with lrelu'), which is also when its boundary rows are
//    published.  The layer-by-layer path reads and re-writes the whole concat gradient in every launch.
// Every spin is bounded (ctx error word).
#include "model.h"

namespace {

// LDS channel-plane stride.  A plane is 5 rows (own three + a halo row either side) x 10 cells (9 + a shared zero column) + frame
// = 56 floats; the stride is 80 = 16 (mod 32) so that the B operand of v_mfma_f32_16x16x4_f32 -- lanes 0..15 read channel k,
// lanes 16..31 channel k + 1 at the same sixteen positions, one ds_read_b32 bank group (bank = dword address mod 32) -- meets
// no bank conflict: the position halves are cut 14 + 13 (cells 0..14 / 15..28 of the own rows: each spans < 16 banks), so the
// two channels of a group occupy disjoint halves of the 32 banks.  (Round 3: stride 56 and halves of 16 + 11 positions -- nine
// of sixteen lanes collided, every B read took four LDS cycles instead of two.)
constexpr int CS = 80;
constexpr int NPOS0 = 14;            // positions of the first half (the second holds 27 - 14 = 13)
#ifndef TFB_NWAVE
#define TFB_NWAVE 8
#endif
constexpr int NWAVE = TFB_NWAVE;     // 8, 16 (four wavefronts per SIMD at <= 128 VGPRs: measured 8 % SLOWER) or 4 (one per SIMD at <= 256
                                     // VGPRs: half of every SIMD's register file stays free for other kernels' wavefronts; conv_layer1
                                     // then has two sub-tiles per wavefront: SK sets of skip values)
constexpr int SK = NWAVE >= 8 ? 1 : 8 / NWAVE;
constexpr int NTHREADS = NWAVE * 64;
constexpr int QU = NWAVE == 16 ? 2 : 4;  // input-channel quads per weight unit
constexpr int AU = QU * 9;           // floats per lane per unit: QU quads x 9 taps
constexpr int BUNIT = AU * 64;       // floats per weight unit
// sub-tiles of conv_layer5 .. conv_layer1's data gradient (2 * output channels / 16); wavefront w owns w, w + NWAVE, ...
__host__ __device__ constexpr int layer_subtiles(int l) { return l == 0 ? 24 : l == 1 ? 20 : l == 2 ? 16 : l == 3 ? 12 : 8; }
__host__ __device__ constexpr int wave_subtiles(int l, int w) {
  return layer_subtiles(l) > w ? (layer_subtiles(l) - w + NWAVE - 1) / NWAVE : 0;
}
__host__ __device__ constexpr int layer_units(int l) { return (l == 0 ? 16 : 8) / QU; }  // units per sub-tile
__host__ __device__ inline int wave_units(int w) {  // weight units of wavefront w per dense block
  int n = 0;
  for (int l = 0; l < 5; ++l) n += wave_subtiles(l, w) * layer_units(l);
  return n;
}
constexpr int SPIN_LIMIT = 1 << 21;
constexpr int P0 = 0, P1 = 64 * CS, Q0 = 128 * CS;  // LDS plane regions: two 64-channel banks (Gout / D[0:64]), D[64:192]
constexpr size_t LDS_BYTES = (size_t)256 * CS * sizeof(float);

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
#ifndef TFB_NACC
#define TFB_NACC 2
#endif
constexpr int NACC = TFB_NACC;  // independent accumulator chains of a sub-tile (summed in the epilogue).  Round 4 measured 1, 2, 3, 4 and the
                                // same loop with in-place inline-asm MFMAs: 1255-1266 us for the K loops alone in every case (in-place: slower)
                                // -- the loop is bound neither by MFMA dependencies nor by the order of its LDS reads (profiles/README.md)

extern __shared__ float lds[];

struct Args {
  const float* wstream;
  const float* gin;        // gradient w.r.t. the output of dense block j1 - 1: (N, *, 81), channels 0..63
  long gin_sn;
  float* dA[TRUNK_FUSED_MAXCAT];         // dA[j], (N, 192, 81)
  const float* cat[TRUNK_FUSED_MAXCAT];  // forward concat buffers (LeakyReLU masks)
  const float* g_a3;       // (N, 64, 81), added at j == 0
  unsigned long long* inbox;
  int* err;
  int* err_dev;
  int nrdb, j0, j1, nimg, img0, epoch;
  float rs, slope;
  unsigned off_xcc;  // XCC_ID table of the handshake (granule offset inside inbox)
  int local_st;
#ifdef TFB_TIMING
  long long* tstamp;   // [block][wave][8] cycle sums (measurement build): 0 sub-tile bodies, 1 barrier waits, 2 layer prologues, 3 halo finish, 7 total
#endif
#ifdef DBM_MEASURE
  int abl;  // libdbm_measure.so only: 1 = no halo exchange, 2 = no epilogue, 4 = every weight unit re-reads the same (cache-hot) address (results are then wrong)
#endif
};

struct Wave {
  int lane, w, t;
  int img, band, cl;
  int nh;           // position half of this wavefront's sub-tiles
  int pos;          // position 0..31 of this lane's column (>= 27: padding)
  int bofs;         // B operand lane base: (lane >> 4) * CS + position offset
  int pofs;         // 11 + position offset: own cell inside a plane
  bool st_ok, up_ok, dn_ok;
  bool local;       // the neighbouring bands run on this XCD: exchange stores stay in its L2
  unsigned me;      // inbox slot of this workgroup
  const float* wp;
#ifdef TFB_TIMING
  long long tsum[8];
#endif
  float skipv[SK][4];   // gradient of the RRDB output at this thread's conv_layer1 outputs (the `x` skip of :402), per sub-tile
};

#define DI __device__ __forceinline__
#ifdef TFB_TIMING
#define TFB_T0() const long long _t0 = (long long)__builtin_amdgcn_s_memtime()
#define TFB_ACC(slot) W.tsum[slot] += (long long)__builtin_amdgcn_s_memtime() - _t0
#else
#define TFB_T0()
#define TFB_ACC(slot)
#endif

DI void issue_unit(float (&A)[AU], const float* p, int lane) {
#pragma unroll
  for (int c = 0; c < AU / 4; ++c) {
    const f4v v = *reinterpret_cast<const f4v*>(p + c * 256 + lane * 4);
    A[4 * c + 0] = v.x; A[4 * c + 1] = v.y; A[4 * c + 2] = v.z; A[4 * c + 3] = v.w;
  }
  if (AU % 4) {
    const f2v v = *reinterpret_cast<const f2v*>(p + (AU / 4) * 256 + lane * 2);
    A[AU - 2] = v.x; A[AU - 1] = v.y;
  }
}

// The QU * 9 MFMAs of one unit.  The B operands of quad q + 1 are requested before the MFMAs of quad q and pinned there
// (hipcc otherwise puts every ds_read right in front of its MFMA: one exposed LDS latency per pair of MFMAs).
DI void mma_unit(const float (&A)[AU], int b, int b_next, float (&bq0)[9], f4v (&acc)[NACC]) {
  // bq0 holds the operands of this unit's first quad (requested during the previous unit); on return those of the next
  // unit's first quad (b_next < 0: there is none)
  float bq[2][9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) bq[0][tap] = bq0[tap];
#pragma unroll
  for (int q = 0; q < QU; ++q) {
    if (q < QU - 1) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) bq[(q + 1) & 1][tap] = lds[b + (q + 1) * 4 * CS + (tap / 3) * 10 + tap % 3];
    } else if (b_next >= 0) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) bq0[tap] = lds[b_next + (tap / 3) * 10 + tap % 3];
    }
    // (round 2) the nine reads above are dealt out BETWEEN the nine MFMAs below -- one MFMA, one DS read, ... -- instead
    // of standing in front of them: their issue slots disappear in the matrix pipe's shadow
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      acc[(q * 9 + tap) % NACC] =
          __builtin_amdgcn_mfma_f32_16x16x4f32(A[q * 9 + tap], bq[q & 1][tap], acc[(q * 9 + tap) % NACC], 0, 0, 0);
    }
#ifndef TFB_NO_INTERLEAVE
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

DI unsigned long long granule_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// local: the reading workgroup runs on this XCD (checked at kernel start, see trunk_fused.hip): the granule stays in its L2
DI void granule_store(unsigned long long* p, float v, unsigned tag, bool local) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if (local) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DI bool xcd_handshake(unsigned long long* table, int base, int mine, unsigned mask, unsigned tag, int lane) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // hwreg(HW_REG_XCC_ID), bits 3:0
  if (threadIdx.x == 0)
    __hip_atomic_store(table + base + mine, ((unsigned long long)tag << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int k = lane & 3;
  const bool partner = (mask >> k) & 1u;
  unsigned long long v = granule_load(table + base + (partner ? k : mine));
  int spins = 0;
#pragma nounroll
  while (partner && (unsigned)(v >> 32) != tag && spins < 4096) {
    __builtin_amdgcn_s_sleep(2);
    v = granule_load(table + base + k);
    ++spins;
  }
  const bool ok = !partner || ((unsigned)(v >> 32) == tag && (unsigned)v == xcc);
  return __all(ok);
}

// Halo rows of NCH freshly finished channels (planes from `plane0`) come from the neighbours' granules.  The loads are
// issued before the wavefront's last sub-tile of the layer (the final block is computed FIRST, so the neighbours have
// usually published by then) and checked after it.
template <int NCH> struct HaloReq {
  static constexpr int R = (NCH * 18 + NTHREADS - 1) / NTHREADS;
  unsigned long long v[R];
};
template <int NCH> DI const unsigned long long* halo_ptr(const Args& a, const Wave& W, int par, int r, int& dst, int plane0) {
  const int g = W.t + NTHREADS * r;
  const int ch = g / 18, side = (g / 9) & 1, c = g % 9;
  const bool have = g < NCH * 18 && (side == 0 ? W.band > 0 : W.band < 2);
  dst = have ? plane0 + ch * CS + (side ? 40 : 0) + c + 1 : -1;
  return a.inbox + ((size_t)W.me * 2 + par) * 2 * 576 + (side * 64 + (have ? ch : 0)) * 9 + c;
}
template <int NCH> DI void halo_issue(const Args& a, const Wave& W, int par, HaloReq<NCH>& q) {
#pragma unroll
  for (int r = 0; r < HaloReq<NCH>::R; ++r) {
    int dst;
    q.v[r] = granule_load(halo_ptr<NCH>(a, W, par, r, dst, 0));
  }
}
template <int NCH> DI void halo_finish(const Args& a, const Wave& W, int plane0, int par, unsigned tag, HaloReq<NCH>& q) {
#pragma unroll
  for (int r = 0; r < HaloReq<NCH>::R; ++r) {
    int dst;
    const unsigned long long* p = halo_ptr<NCH>(a, W, par, r, dst, plane0);
    if (dst >= 0) {
      unsigned long long v = q.v[r];
      int spins = 0;
#pragma nounroll
      while ((unsigned)(v >> 32) != tag) {
        __builtin_amdgcn_s_sleep(1);
        v = granule_load(p);
        if (++spins > SPIN_LIMIT) { *a.err = 1; *a.err_dev = 1; break; }
      }
      lds[dst] = __uint_as_float((unsigned)v);
    }
  }
}

// One sub-tile (16 output channels x 16 positions) of one layer's data gradient.
//  KL: 4 = conv_layer5 (input Gout, 64 channels, NU = 4 units), 3..0 = conv_layer4..1 (32 input channels, NU = 2)
#ifndef TFB_LATE_LOADS
#define TFB_LATE_LOADS 0   // (measured: chain 1.52 ms with, 1.49-1.50 without -- the stall only moves into the K loop; kept for A/B)
#endif
// TFB_LATE_LOADS (round 4): the first memory instruction a wavefront issues behind a sub-tile's epilogue stores stalls until they
// have drained (trunk_fused.hip, TF_TIMING).  A sub-tile therefore issues NOTHING before its first unit's MFMAs: the weights of
// its second unit were requested before the previous sub-tile's epilogue (into the buffer that one's last unit had finished
// with), and the epilogue's own global reads (LeakyReLU masks) and the layer's halo requests follow the first unit.
template <int KL, int NCH>
DI void sub_tile(const Args& a, Wave& W, float (&A0)[AU], float (&A1)[AU], int s, int j, int gin_region, int dlow_region,
                 int serial, const float (&bfirst)[9], HaloReq<NCH>* hq) {
  constexpr int NU = (KL == 4 ? 16 : 8) / QU;
  constexpr int fin0 = KL == 4 ? 160 : KL == 3 ? 128 : KL == 2 ? 96 : KL == 1 ? 64 : 0;  // first channel that is final
  const int lane = W.lane;
  const int mt = (W.w >> 1) + (NWAVE / 2) * s;  // 16-channel output tile
  const int ch0 = 16 * mt + 4 * (lane >> 4);   // this lane's four output channels ch0 .. ch0 + 3
  const bool fin = 16 * mt >= fin0;            // wave-uniform
  const bool third = (j % 3 == 2), first = (j % 3 == 0);
  // B operand planes: Gout bank (conv5) or D[64 + 32 KL ...]
  const int breg = (KL == 4 ? gin_region : Q0 + 32 * KL * CS) + W.bofs;

  float maskv[4] = {1.f, 1.f, 1.f, 1.f};
  const bool use_mask = fin && (KL > 0 || j == 0);
  const unsigned gofs = (unsigned)((W.img * 192 + ch0) * 81 + W.band * 27 + W.pos);

  f4v acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
  float bq0[9];  // B operands of the next quad to be multiplied; every sub-tile of a layer starts with the same nine
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) bq0[tap] = bfirst[tap];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    float (&cur)[AU] = (u & 1) ? A1 : A0;
    float (&nxt)[AU] = (u & 1) ? A0 : A1;
#pragma unroll
    for (int i = 0; i < AU; ++i) asm volatile("" ::"v"(cur[i]));  // the wait for this unit's weights, BEFORE the next issue
    __builtin_amdgcn_sched_barrier(0);
    auto epilogue_reads = [&]() {  // what the epilogue reads from global memory: masks of final channels
      if (W.st_ok && use_mask) {
        const float* C = a.cat[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) maskv[r] = C[gofs + r * 81];
      }
    };
    if (!TFB_LATE_LOADS && u == 0) epilogue_reads();
    if (!(TFB_LATE_LOADS && u == 0)) {
      issue_unit(nxt, W.wp, lane);
      if (!DBM_ABL_BIT(a, 4)) W.wp += BUNIT;  // (abl 4: every unit re-reads the same weights -- always cache-hot; results wrong)
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_unit(cur, breg + u * QU * 4 * CS, u + 1 < NU ? breg + (u + 1) * QU * 4 * CS : -1, bq0, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (TFB_LATE_LOADS && u == 0) {
      epilogue_reads();
      if (hq) halo_issue<NCH>(a, W, serial & 1, *hq);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

#ifndef TFB_NO_EARLY_WAIT
  {
    // The next sub-tile's first weights (requested during the last unit above) are awaited HERE, before the epilogue's global
    // and granule stores: see trunk_fused.hip (vmcnt counts the stores too, and behind their branches hipcc waits with vmcnt(0)).
    float (&pend)[AU] = ((NU - 1) & 1) ? A0 : A1;
#pragma unroll
    for (int i = 0; i < AU; ++i) asm volatile("" ::"v"(pend[i]));
    __builtin_amdgcn_sched_barrier(0);
    if (TFB_LATE_LOADS) {  // the next sub-tile's SECOND unit, into the buffer this sub-tile's last unit has finished with
      static_assert(NU % 2 == 0, "every sub-tile starts on buffer A0");
      issue_unit(A1, W.wp, lane);
      if (!DBM_ABL_BIT(a, 4)) W.wp += BUNIT;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#endif
  // ---- epilogue: this lane's four cells ----
  if (DBM_ABL_BIT(a, 2)) {
#pragma unroll
    for (int c = 0; c < NACC; ++c) asm volatile("" ::"v"(acc[c]));
    return;
  }
  const float sc = third ? a.rs * a.rs : a.rs;
  const float r1s = third ? a.rs : 1.f;
  const unsigned tag_out = ((unsigned)a.epoch << 12) | (unsigned)(serial + 1);
  if (W.st_ok) {
    float* gdst = a.dA[j];
    unsigned long long* obox = a.inbox + ((serial & 1) ? 2 * 576 : 0);
    const bool publish = fin && !(KL == 0 && j == a.j0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = ch0 + r;
      const int cell = (ch < 64 ? dlow_region + ch * CS : Q0 + (ch - 64) * CS) + W.pofs;
      float v = acc[0][r];
#pragma unroll
      for (int c = 1; c < NACC; ++c) v += acc[c][r];
      if (KL == 4) {
        v *= sc;
        if (ch < 64) v += r1s * lds[gin_region + ch * CS + W.pofs];  // d out / d a0  (:358, :402)
      } else {
        if (KL == 0) {
          if (first) v += W.skipv[SK > 1 ? s : 0][r];          // d (RRDB out) / d x  (:402)
          if (j == 0) v += a.g_a3[(unsigned)((W.img * 64 + ch) * 81 + W.band * 27 + W.pos)];  // a3 = a1 + ...  (:551)
        }
        v += lds[cell];
      }
      if (use_mask) v = maskv[r] >= 0.f ? v : a.slope * v;
      lds[cell] = v;
      if (KL == 0 && first) W.skipv[SK > 1 ? s : 0][r] = v;
      if (fin) {
        gdst[gofs + r * 81] = v;
        if (publish) {
          const int pch = ch - fin0;
          if (W.up_ok) granule_store(obox + ((size_t)(W.me - 1) * 4 + 1) * 576 + pch * 9 + W.pos, v, tag_out, W.local);
          if (W.dn_ok) granule_store(obox + ((size_t)(W.me + 1) * 4 + 0) * 576 + pch * 9 + (W.pos - 18), v, tag_out, W.local);
        }
      }
    }
  }
}

template <int KL> DI void layer(const Args& a, Wave& W, float (&A0)[AU], float (&A1)[AU], int j, int gin_region, int dlow_region,
                               int serial) {
  // sub-tiles of this layer: 2 * (64 + 32 KL) / 16 = 24, 20, 16, 12, 8; wavefront w owns w, w + 8, w + 16
  constexpr int S = (64 + 32 * KL) / 8;
  constexpr int SMAX = (S + NWAVE - 1) / NWAVE;
  // keep hipcc from hoisting every address of every (layer, sub-tile, register) out of the dense-block loop (it spills)
  asm volatile("" : "+v"(W.pos), "+v"(W.bofs), "+v"(W.pofs), "+v"(W.lane));
  // the block that becomes final in this layer feeds the next one: its sub-tiles (the highest channels) go first
  constexpr int NCH = KL == 0 ? 64 : 32;
  const int plane0 = KL == 0 ? dlow_region : Q0 + 32 * (KL - 1) * CS;
  const bool fetch = !(KL == 0 && j == a.j0) && !DBM_ABL_BIT(a, 1);
  HaloReq<NCH> hq;
  // the first quad's nine B operands are the same for all of this wavefront's sub-tiles of the layer: requested once
  float bfirst[9];
  {
    TFB_T0();
    const int b0 = (KL == 4 ? gin_region : Q0 + 32 * KL * CS) + W.bofs;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) bfirst[tap] = lds[b0 + (tap / 3) * 10 + tap % 3];
#ifdef TFB_TIMING
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) asm volatile("" ::"v"(bfirst[tap]));
#endif
    TFB_ACC(2);
  }
#pragma unroll
  for (int s = SMAX - 1; s >= 0; --s) {
    // (the halo requests go out before the wavefront's LAST sub-tile of the layer -- the final block is computed first, so the
    //  neighbours have usually published by then; TFB_LATE_LOADS: behind that sub-tile's first unit)
    const bool mine = W.w + NWAVE * s < S;
    if (s == 0 && fetch && !(TFB_LATE_LOADS && mine)) {
      halo_issue<NCH>(a, W, serial & 1, hq);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (mine) {
      TFB_T0();
      sub_tile<KL, NCH>(a, W, A0, A1, s, j, gin_region, dlow_region, serial, bfirst, (s == 0 && fetch) ? &hq : nullptr);
      TFB_ACC(0);
    }
  }
  if (fetch) {
    TFB_T0();
    halo_finish<NCH>(a, W, plane0, serial & 1, ((unsigned)a.epoch << 12) | (unsigned)(serial + 1), hq);
    TFB_ACC(3);
  }
  {
    TFB_T0();
    __syncthreads();
    TFB_ACC(1);
  }
}

}  // namespace

__global__ __launch_bounds__(NTHREADS, NWAVE == 4 ? 2 : 1) void trunk_fused_bwd_kernel(Args a) {
  Wave W;
  W.t = threadIdx.x; W.lane = W.t & 63; W.w = W.t >> 6;
  const int B = blockIdx.x;
  W.cl = (B / 24) * 8 + (B % 8);
  W.band = (B / 8) % 3;
  if (W.cl >= a.nimg) return;
  W.img = a.img0 + W.cl;
  W.me = (unsigned)(W.cl * 3 + W.band);
  W.local = a.local_st && xcd_handshake(a.inbox + a.off_xcc, 4 * W.cl, W.band,
                                        (W.band > 0 ? 1u << (W.band - 1) : 0u) | (W.band < 2 ? 1u << (W.band + 1) : 0u),
                                        ((unsigned)a.epoch << 12) | 0xFFFu, W.lane);
  W.nh = W.w & 1;
  {
    const int l15 = W.lane & 15;
    W.st_ok = W.nh == 0 ? l15 < NPOS0 : l15 < 27 - NPOS0;
    // padding lanes compute on their half's first position (a broadcast read); their results go nowhere
    const int n = W.nh * NPOS0 + (W.st_ok ? l15 : 0);
    W.pos = n;
    const int po = (n / 9) * 10 + n % 9;
    W.bofs = (W.lane >> 4) * CS + po;
    W.pofs = 11 + po;
    W.up_ok = W.st_ok && n < 9 && W.band > 0;
    W.dn_ok = W.st_ok && n >= 18 && W.band < 2;
  }
  {
    const int nblk = a.nrdb - a.j1;  // dense blocks already done by earlier launches
    // wavefronts 2k and 2k + 1 compute the two position halves of the same output channels: same weights, so ONE stream
    // per pair (the second reader hits in L1 / merges with the first's miss: chain 1.82 -> 1.69 ms)
    size_t before = 0;
    for (int q = 0; q < (W.w >> 1); ++q) before += wave_units(2 * q);
    W.wp = a.wstream + ((size_t)a.nrdb * before + (size_t)nblk * wave_units(W.w)) * BUNIT;
  }
  for (int i = W.t; i < 256 * CS; i += NTHREADS) lds[i] = 0.f;
  __syncthreads();
  // Gout of the first dense block of this launch: channels 0..63, rows 3*band-1 .. 3*band+3, into bank P0
  const float* gin = a.gin + (size_t)W.img * a.gin_sn;
  for (int i = W.t; i < 64 * 45; i += NTHREADS) {
    const int ch = i / 45, rc = i % 45, r = rc / 9, c = rc % 9;
    const int row = 3 * W.band - 1 + r;
    if (row >= 0 && row < 9) lds[P0 + ch * CS + r * 10 + c + 1] = gin[ch * 81 + row * 9 + c];
  }
  {  // launches cover whole RRDBs: the skip source of the first RRDB is the launch's Gout itself
#pragma unroll
    for (int k = 0; k < SK; ++k) {
      const int ch0 = 16 * ((W.w >> 1) + (NWAVE / 2) * k) + 4 * (W.lane >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)  // (only the wavefronts that own conv_layer1 sub-tiles: channels < 64)
        W.skipv[k][r] = (W.st_ok && ch0 < 64) ? gin[(ch0 + r) * 81 + W.band * 27 + W.pos] : 0.f;
    }
  }
  __syncthreads();

#ifdef TFB_TIMING
  for (int i = 0; i < 8; ++i) W.tsum[i] = 0;
  const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
#ifdef TF_PRIO
  // the second wavefront of every SIMD (dispatched later = the loser of every issue arbitration by age) gets a static priority
  if (W.w >= NWAVE / 2) __builtin_amdgcn_s_setprio(TF_PRIO);
#endif
#ifdef TFB_PRIO_ALL
  __builtin_amdgcn_s_setprio(TFB_PRIO_ALL);   // (measurement: every wavefront of the chain above the co-resident kernels' wavefronts)
#endif
  float A0[AU], A1[AU];
  issue_unit(A0, W.wp, W.lane);
  W.wp += BUNIT;
  if (TFB_LATE_LOADS) {
    issue_unit(A1, W.wp, W.lane);
    W.wp += BUNIT;
  }
  int serial = 0;
  int par = 0;  // Gout bank
  for (int j = a.j1 - 1; j >= a.j0; --j) {
    const int gin_region = par ? P1 : P0, dlow_region = par ? P0 : P1;
    layer<4>(a, W, A0, A1, j, gin_region, dlow_region, serial + 0);
    layer<3>(a, W, A0, A1, j, gin_region, dlow_region, serial + 1);
    layer<2>(a, W, A0, A1, j, gin_region, dlow_region, serial + 2);
    layer<1>(a, W, A0, A1, j, gin_region, dlow_region, serial + 3);
    layer<0>(a, W, A0, A1, j, gin_region, dlow_region, serial + 4);
    serial += 5;
    par ^= 1;
  }
#ifdef TFB_TIMING
  W.tsum[7] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
  if (W.lane == 0 && a.tstamp)
    for (int i = 0; i < 8; ++i) a.tstamp[((size_t)blockIdx.x * NWAVE + W.w) * 8 + i] = W.tsum[i];
#endif
}

// dst-driven gather of the transposed, tap-flipped trunk weights into the per-wavefront streams
constexpr int NPAIR = NWAVE / 2;  // wavefronts 2k, 2k + 1 share a stream
struct PackTab {
  long base[NPAIR + 1];   // first float of pair k's stream
  int upr[NPAIR];         // its units per dense block
  int cum[NPAIR][6];      // ... and where conv_layer5 .. conv_layer1 start inside them
  int nsub[NPAIR][5];
};
// One wavefront per weight unit: a lane's AU values are QU runs of nine consecutive floats of the OIHW tensor (its gradient
// output channel = forward input channel, forward output channels 4 (QU u + q) + (lane >> 4), taps reversed), written as
// the 16-byte pieces issue_unit reads back (the element-wise gather this replaces: 47 us).
__global__ __launch_bounds__(256) void pack_trunk_fused_bwd_kernel(const float* const* wsrc, float* wstream, int nrdb, PackTab tab,
                                                                   int total_units) {
  static_assert(AU % 4 == 0, "unit pieces are 16 bytes");
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= total_units) return;
  int w = 0;  // pair index: pair k's units start at tab.base[k] / BUNIT
#pragma unroll
  for (int k = 1; k < NPAIR; ++k) w += (long)task * BUNIT >= tab.base[k] ? 1 : 0;
  const int rem = task - (int)(tab.base[w] / BUNIT);
  const int upr = tab.upr[w];
  const int rblk = rem / upr, ui = rem % upr;              // dense blocks in processing order: j = nrdb - 1 - rblk
  const int j = nrdb - 1 - rblk;
  // unit -> (layer, sub-tile s, unit u); the wavefront walks its sub-tiles from the highest channels down
  int l = 0;
#pragma unroll
  for (int k = 1; k < 5; ++k) l += ui >= tab.cum[w][k] ? 1 : 0;
  const int KL = 4 - l, nu = layer_units(l), ul = ui - tab.cum[w][l];
  const int s = tab.nsub[w][l] - 1 - ul / nu, u = ul % nu;
  const int mt = w + (NWAVE / 2) * s;
  const int ci = 16 * mt + (lane & 15);                   // forward input channel = gradient output channel
  const int Cin = 64 + 32 * KL;
  const float* src = wsrc[j * 5 + KL] + ((long)(4 * QU * u + (lane >> 4)) * Cin + ci) * 9;
  float A[AU];
#pragma unroll
  for (int q = 0; q < QU; ++q)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) A[q * 9 + tap] = src[(long)q * 4 * Cin * 9 + (8 - tap)];
  float* dst = wstream + (size_t)task * BUNIT;
#pragma unroll
  for (int c = 0; c < AU / 4; ++c)
    *reinterpret_cast<f4v*>(dst + c * 256 + lane * 4) = (f4v){A[4 * c], A[4 * c + 1], A[4 * c + 2], A[4 * c + 3]};
}

// ------------------------------------------------------------------------------------------------------------------
size_t trunk_fused_bwd_stream_floats(int nrdb) {
  size_t units = 0;
  for (int q = 0; q < NWAVE / 2; ++q) units += wave_units(2 * q);
  return (size_t)nrdb * units * BUNIT + 8 * BUNIT;
}

void launch_pack_trunk_fused_bwd(const float* const* d_wsrc, float* wstream, int nrdb, hipStream_t s) {
  PackTab tab;
  long base = 0;
  for (int q = 0; q < NPAIR; ++q) {
    const int w = 2 * q;
    tab.base[q] = base;
    tab.upr[q] = wave_units(w);
    int c = 0;
    for (int l = 0; l < 5; ++l) {
      tab.cum[q][l] = c;
      tab.nsub[q][l] = wave_subtiles(l, w);
      c += wave_subtiles(l, w) * layer_units(l);
    }
    tab.cum[q][5] = c;
    base += (long)nrdb * tab.upr[q] * BUNIT;
  }
  tab.base[NPAIR] = base;
  const int total_units = (int)(base / BUNIT);
  hipLaunchKernelGGL(pack_trunk_fused_bwd_kernel, dim3((total_units + 3) / 4), dim3(256), 0, s, d_wsrc, wstream, nrdb, tab, total_units);
  DBM_HIP(hipGetLastError());
}

void launch_trunk_fused_bwd(const TrunkFusedBwdLaunch& L, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    DBM_HIP(hipFuncSetAttribute((const void*)trunk_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr = true;
  }
  DBM_CHECK(L.nrdb + 1 <= TRUNK_FUSED_MAXCAT, "fused trunk: too many dense blocks");
  DBM_CHECK(L.j0 % 3 == 0 && L.j1 % 3 == 0 && L.j0 < L.j1 && L.j1 <= L.nrdb, "fused trunk backward: launches cover whole RRDBs");
  Args a;
  a.wstream = L.wstream; a.gin = L.gin; a.gin_sn = L.gin_sn; a.g_a3 = L.g_a3; a.inbox = L.inbox; a.err = L.err; a.err_dev = L.err_dev;
  for (int i = 0; i < TRUNK_FUSED_MAXCAT; ++i) {
    a.dA[i] = i < L.nrdb ? L.dA[i] : nullptr;
    a.cat[i] = i < L.nrdb ? L.cat[i] : nullptr;
  }
  a.nrdb = L.nrdb; a.j0 = L.j0; a.j1 = L.j1; a.nimg = L.nimg; a.img0 = L.img0; a.epoch = L.epoch & 0xFFFFF;
  a.rs = L.rs; a.slope = L.slope;
#ifdef DBM_MEASURE
  static const int abl = DBM_MEASURE_ENV("TFB_ABL");
  a.abl = abl;
#endif
  a.off_xcc = (unsigned)trunk_fused_xcc_offset(64);
  a.local_st = trunk_local_stores();
  const int grid = ((L.nimg + 7) / 8) * 24;
  if (g_profiler.enabled) {
    // algorithmic bytes: the flipped weights of the launch's dense blocks once; per image the incoming gradient (64 channels),
    // every block's stored concatenation (192: the LeakyReLU masks) in, every block's conv-output gradients dA (192) out
    const double nb = L.j1 - L.j0;
    const double bytes = 4.0 * 26624.0 * 9.0 * nb + 4.0 * 81.0 * L.nimg * (64.0 + 64.0 + 2.0 * 192.0 * nb);
    char tag[40];
    snprintf(tag, sizeof(tag), "trunk_bwd_%d..%d_n%d", L.j0, L.j1, L.nimg);
    g_profiler.begin(s, 3, 2.0 * 19408896.0 * (L.j1 - L.j0) * L.nimg, bytes, tag, grid);
  }
#ifdef TFB_TIMING
  static long long* d_ts = nullptr;
  if (!d_ts) DBM_HIP(hipMalloc((void**)&d_ts, sizeof(long long) * 256 * NWAVE * 8));
  a.tstamp = d_ts;
#endif
  hipLaunchKernelGGL(trunk_fused_bwd_kernel, dim3(grid), dim3(NTHREADS), LDS_BYTES, s, a);
  if (g_profiler.enabled) g_profiler.end(s);
#ifdef TFB_TIMING
  if (g_profiler.enabled && g_profiler.serial) {   // (the serialised profile pass of tools/experiments/step_shapes.py)
    std::vector<long long> h((size_t)grid * NWAVE * 8);
    DBM_HIP(hipDeviceSynchronize());
    DBM_HIP(hipMemcpy(h.data(), d_ts, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int b : {0, 1, 100}) {
      if (b >= grid) continue;
      for (int w = 0; w < NWAVE; ++w) {
        const long long* t = &h[((size_t)b * NWAVE + w) * 8];
        fprintf(stderr, "chain timing block %3d wave %d: subtiles %9lld  barrier %9lld  prologue %8lld  halo %8lld  total %9lld cycles (s_memtime)\n", b, w,
                t[0], t[1], t[2], t[3], t[7]);
      }
    }
  }
#endif
  DBM_HIP(hipGetLastError());
}
