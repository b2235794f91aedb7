// The whole RRDB trunk of the generator forward (srgan_train.py:333-360 ResidualDenseBlock.forward, :393-404
// ResInResDenseBlock.forward, :546 residual_network) on 9x9 planes as ONE persistent launch.
//
// Layer by layer the 9x9 stage is a chain of 5 * 3 * n_rrdb launches of 2-12 us of matrix work each; a kernel boundary
// plus prologue / epilogue costs more than the MFMAs between two of them.  Here a workgroup owns TP consecutive positions of
// the launch's images laid end to end (image-major, row-major) for the whole trunk -- one 32-wide MFMA column tile:
//   TP = 27: three rows of one image (tiles never cross an image; 27 of the 32 columns work, 3 workgroups per image);
//   TP = 32: every column works; 64 images = 5184 positions = 162 workgroups instead of 192 (a tile starts anywhere in a
//            row and may run from the end of one image into the next: the LDS planes hold the tile's rows in order with
//            ONE zero row between the two images, which is the bottom padding of the first and the top padding of the second):
//  * the dense block's concat (192 channels x (own + halo rows)) lives in LDS, zero-framed, so the B operand of every
//    tap is one ds_read_b32 with an immediate offset and no border logic;
//  * the K axis (input channels) is split over eight wavefronts in units of four channels; channel quad q belongs to
//    wavefront q % 8 in every layer, so a wavefront is the only reader of "its" planes and fetches their halo rows alone;
//  * weights stream from L2 in the order each wavefront consumes them (pack_trunk_fused_kernel), 16 bytes per lane per
//    load, one unit (4 channels x 9 taps = 18 MFMAs) ahead, across layer and dense-block boundaries;
//  * neighbouring tiles exchange the boundary of every layer's output -- a 3x3 tap of an own position reaches at most ten
//    positions before the tile's first and ten after its last one -- through data-tagged 8-byte granules ({value, layer
//    tag}, one relaxed agent-scope store / load each: no fences, no flags).  A layer's K loop runs the old channels and
//    the newest channels' halo-free taps first (TP = 27: the middle kernel row; TP = 32: the centre tap);
//  * split-K partial tiles are reduced through LDS; the epilogue (bias, LeakyReLU, `a5*rs + a0`, `a3*rs + x`) writes the
//    LDS planes, the global concat buffers the backward pass reads (training) and the neighbours' granules.
//  * helper mode (HM, TP = 27, passes that keep nothing): a fourth workgroup per image computes output channels 32..63 of
//    every conv_layer5 for the three bands (helper_trunk), which hand it every layer output and take its channels back
//    through the same kind of granules: 360 instead of 468 MFMAs per wavefront and dense block.
// Every spin is bounded (an error word is raised instead of a hang); workgroups never wait for anything but the two
// neighbouring tiles (and their image's helper).  <= 248 VGPRs: one workgroup per CU, a 64-image launch (192 / 162 / 256
// workgroups) is resident at once on the 256 CUs; kernels of other streams only delay it (they finish), and persistent
// launches are serialised among themselves (dbm_ctx::persist_begin).
#include "model.h"

namespace {

constexpr int HS = 10;           // halo slots per side (positions before the first / after the last own position)
constexpr int NWAVE = 8;
constexpr int NTHREADS = NWAVE * 64;
constexpr int UNIT = 1152;       // floats of one (unit, out-channel tile) weight block: 64 lanes x 18
constexpr int WAVE_RDB = 26 * UNIT;  // (2+3+4+5) + 2*6 blocks per wavefront per dense block
constexpr int SPIN_LIMIT = 1 << 21;

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

extern __shared__ float lds[];   // act[192][CS] | red[8][16][64]
// floats per LDS channel plane: (own rows + a halo row either side [+ the zero row between two images]) x (9 + 1 shared
// zero column) + frame
template <int TP> struct Geo {
  static constexpr int CS = TP == 27 ? 56 : 88;
  static constexpr int RED0 = 192 * CS;
  static constexpr size_t LDS_BYTES = (size_t)(RED0 + NWAVE * 16 * 64) * sizeof(float);
};

struct Args {
  const float* wstream;   // [wave][rdb][26][UNIT]: every wavefront reads one contiguous stream
  const float* bstream;   // [rdb][192]: conv1..4 (32 each), conv5 (64)
  const float* in;        // (N, 192, 81) concat buffer of dense block 0; channels 0..63 hold the trunk input
  float* cat[TRUNK_FUSED_MAXCAT];  // training: nrdb + 1 concat buffers (cat[0] == in), in the kernel arguments: a
                          // pointer table in global memory costs a dependent vector load per epilogue
  int store_all;
  float* out;             // inference: concat buffer receiving the trunk output in channels 0..63
  unsigned long long* inbox;  // [tiles][2 parities][2 sides][64][HS] granules {value, tag}
  int* err;
  int* err_dev;
  int nrdb, nimg, img0, epoch;
  int ntiles, tpx;         // tiles of this launch; tiles per XCD (block b = tile (b % 8) * tpx + b / 8)
  unsigned off_xcc;        // granule offset of the XCC_ID table [image][4] (xcd_handshake)
  int local_st;            // exchange stores may stay in the XCD's L2 when the reader is on the same XCD (DBM_TRUNK_LOCAL_ST)
#ifdef TF_TIMING
  long long* tstamp;       // [block][wave][8] cycle sums (measurement build): 0 K loops, 1 first barrier, 2 reduction + epilogue, 3 last barrier,
                           // 4 layer prologue (weights wait + granule requests), 7 total
#endif
  unsigned off_hb, off_bh; // helper mode: granule offsets of the helpers' inboxes [image][2][5][32][81] and of the boxes the
                           // helpers fill for their bands [image][2][32][81]
  float rs, slope;
};

struct Wave {
  int lane, w, t;
  int bofs;                // lane's B base: (lane >> 5) * CS + position offset
  int n;                   // MFMA column 0..31 (padding lane unless st_ok)
  int pofs;                // own cell inside a plane (valid lanes)
  const float* wp;         // next weight block of this wavefront
  float xres[4];           // RRDB input at this thread's four conv5 outputs
  // epilogue constants of this thread's first output (register r = 2 w, tile 0); the others add immediates
  unsigned ep_l;           // LDS cell of (channel m0, own position)
  unsigned ep_g;           // element offset inside a concat buffer: (img * 192 + m0) * 81 + position inside the image
  unsigned ep_up, ep_dn;   // granule index inside the previous / next tile's inbox (parity 0, channel m0)
  unsigned hl_g[2];        // halo fetch: granule index of slot s = lane (+64) inside my inbox (parity 0, channel 0)
  int hl_l[2];             // ... and its LDS cell relative to the quad's first plane (-1: nothing to fetch)
  // helper mode: the 4 x 45 values (own rows + the row above / below) of a quad the helper computed -- granule index
  // inside the helper's box of this image (channel 0) and LDS cell, for slot s = lane + 64 r; what this thread publishes
  // to the helper: (channel m0) * 81 + position inside the image
  int cl;
  unsigned hh_g[3];
  int hh_l[3];
  unsigned ep_h;
  bool st_ok, up_ok, dn_ok;
  bool local;              // every workgroup that reads this one's granules runs on this XCD
#ifdef TF_TIMING
  long long tsum[8];
#endif
};
#ifdef TF_TIMING
#define TF_NOW() ((long long)__builtin_amdgcn_s_memtime())
#define TF_LAP(slot) do { const long long _n = TF_NOW(); W.tsum[slot] += _n - _tl; _tl = _n; } while (0)
#else
#define TF_LAP(slot)
#endif

#define DI __device__ __forceinline__

#define C5U(s) ((s) == 0 ? 1 : (s) == 1 ? 0 : (s))  // conv_layer5: unit consumed at step s
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

// SEL 0: all nine taps; 1: the taps that reach no halo cell (TP = 27: the middle kernel row -- tiles are whole rows; TP = 32:
// the centre tap); 2: the others
template <int TP, int NM, int SEL> DI void mma_taps(const float (&A)[36], int b, f16v (&acc)[2]) {
  constexpr int CS = Geo<TP>::CS;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const bool halo_free = TP == 27 ? tap / 3 == 1 : tap == 4;
    if (SEL == 1 && !halo_free) continue;
    if (SEL == 2 && halo_free) continue;
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
// local: the reader's workgroup runs on the SAME XCD as this one (xcd_handshake): the granule may stay in that XCD's L2 --
// a store that is write-through only as far as the L2 (`sc0`; the reader's agent-scope load finds the dirty line there) instead
// of an agent-scope store, which gfx950 writes through to the fabric (`sc1`): the exchange then costs no memory-side write
// traffic while the line stays there (PMC, 64-image pass in helper form: WRITE_SIZE 457 -> 298 MB, FETCH_SIZE x 2 868 -> 412 MB -- the
// weight stream evicts part of the dirty granules, 1.6 MB per dense block and XCD against a 4 MB L2 --; retained pass 739 -> 451 MB,
// backward chain 856 -> 550 MB: profiles/r3/n_local_stores_traffic_pmc.json) and the passes are 5 % shorter.
// The tag travels with the value, so a granule that did NOT become visible can only delay its reader (bounded spin ->
// status 7), never feed it a wrong value.
DI void granule_store(unsigned long long* p, float v, unsigned tag, bool local) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if (local) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Which XCD am I on, and are the workgroups I exchange granules with on the same one?  Every workgroup publishes its XCC_ID
// (agent scope, tagged with the launch's epoch) in slot `mine` of the table and reads the slots in `mask` (bit k = slot
// base + k).  Workgroups are dealt to the XCDs round-robin by index and the tile mapping puts an image's workgroups on one
// XCD, so the answer is normally yes; it is CHECKED because a "no" with local stores would be a stall.  A partner that
// does not show up within the spin bound just means agent-scope stores.
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

}  // namespace

#ifndef TF_LATE_LOADS
#define TF_LATE_LOADS 1
#endif
// The request a layer's FIRST unit makes for its second unit's weights (layer KN = 0..4), factored out so that the previous
// layer can issue it before its epilogue (TF_LATE_LOADS).  Same stream bookkeeping as in dense_layer's loop at u = 0.
template <int KN, bool HM> DI void issue_second_unit(float (&buf)[36], Wave& W, int lane) {
  if (!HM) {
    if (KN == 4) { issue_loads<2>(buf, W.wp, lane); W.wp += 2 * UNIT; }
    else { issue_loads<1>(buf, W.wp, lane); W.wp += UNIT; }
  } else {
    if (KN == 4) issue_loads<1>(buf, W.wp + 2 * C5U(1) * UNIT, lane);   // (conv_layer5 walks its units in the order C5U; W.wp stays)
    else { issue_loads<1>(buf, W.wp, lane); W.wp += UNIT; }
  }
}

// One layer of a dense block.  K = 0..4 (conv_layer1..5).  A0 / A1: the weight ping-pong (static parity per dense block).
// HM (helper mode, TP = 27): a fourth workgroup per image computes output channels 32..63 of every conv_layer5 for the three
// bands (helper_trunk below); a band computes channels 0..31, hands every output of every layer to the helper and takes the
// helper's channels -- own rows and halo rows -- from the helper's box at the next dense block's conv_layer1.
template <int TP, int K, bool HM>
__device__ __forceinline__ void dense_layer(const Args& a, Wave& W, float (&A0)[36], float (&A1)[36], int j, bool last_rdb) {
  constexpr int CS = Geo<TP>::CS, RED0 = Geo<TP>::RED0;
  constexpr int NM = (K == 4 && !HM) ? 2 : 1;
  constexpr int U = 2 + K;
  constexpr int base = K == 0 ? 0 : K == 1 ? 2 : K == 2 ? 5 : K == 3 ? 9 : 14;
  const int lane = W.lane, w = W.w;
#ifdef TF_TIMING
  long long _tl = TF_NOW();
#endif
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
  // granule slot s (0 .. 8 HS - 1) of a quad: channel e = s / (2 HS), side = (s / HS) & 1 (0: before, 1: after), position s % HS
  constexpr int NQ = (K == 0 && !HM) ? 2 : 1;        // newest quads of this wavefront whose halo comes from the neighbours
#ifdef TF_TIMING
  __builtin_amdgcn_sched_barrier(0);
  TF_LAP(5);   // (slot 5: the layer's first VALU work, before any memory instruction)
  __builtin_amdgcn_sched_barrier(0);
#endif
  const unsigned long long* gp[NQ][2];
  unsigned long long gv[NQ][2];
  int gdst[NQ][2];
  float bias[2 * NM];
  // Round 4: the first memory instruction a wavefront issues behind the previous layer's epilogue stores (granule stores are
  // write-through) stalls until those have drained -- 1 700 cycles per layer for every wavefront but the first to arrive
  // (TF_TIMING: 11 % of the launch).  Layers whose halo is not needed at once (conv_layer2..5: the newest unit is their last)
  // therefore issue NOTHING at their start: these requests follow the first unit's MFMAs, and the weights of the layer's second
  // unit were requested before the previous layer's epilogue (below).
  constexpr bool LATE = TF_LATE_LOADS && K > 0;
  auto layer_requests = [&]() {
    const unsigned long long* inb = a.inbox + (par_in ? 2 * 64 * HS : 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int quad = K == 0 ? w + 8 * q : w + 8 * (1 + K);
      const int pch = K == 0 ? 4 * (w + 8 * q) : 4 * w;  // channel of the producing layer's output
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        gdst[q][r] = (need_halo && W.hl_l[r] >= 0) ? W.hl_l[r] + quad * 4 * CS : -1;
        gp[q][r] = inb + (W.hl_g[r] + pch * HS);
        gv[q][r] = granule_load(gp[q][r]);
      }
    }
#pragma unroll
    for (int k = 0; k < 2 * NM; ++k) {
      const int r = 2 * w + (k & 1);
      bias[k] = a.bstream[j * 192 + (K < 4 ? 32 * K : 128) + (k >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    }
  };
  if (!LATE) layer_requests();
  // helper mode, conv_layer1: channels 32 + 4 w .. of the block input come from the helper
  const bool from_helper = HM && K == 0 && j > 0;
  const unsigned long long* hp[3];
  unsigned long long hv[3];
  if (HM && K == 0) {
    const unsigned long long* bh = a.inbox + a.off_bh + ((size_t)W.cl * 2 + ((j + 1) & 1)) * (32 * 81) + 4 * w * 81;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      hp[r] = bh + W.hh_g[r];
      hv[r] = granule_load(hp[r]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  f16v acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  TF_LAP(4);

#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool par = ((base + u) & 1) != 0;
    float (&cur)[36] = par ? A1 : A0;
    float (&nxt)[36] = par ? A0 : A1;
    // the next unit of this wavefront's stream (next layer / next dense block included; the stream is padded at its end)
    const bool next_is_c5 = (u + 1 < U) ? (K == 4) : (K == 3);   // the next unit belongs to conv_layer5 (two tiles)
    if (u > 0) {
#pragma unroll
      for (int i = 0; i < 18 * NM; ++i) asm volatile("" ::"v"(cur[i]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TF_LATE_LOADS && u == 0) {
      // (this unit's successor was requested before the previous layer's epilogue: issue_second_unit)
    } else if (!HM) {
      if (next_is_c5) { issue_loads<2>(nxt, W.wp, lane); W.wp += 2 * UNIT; }
      else { issue_loads<1>(nxt, W.wp, lane); W.wp += UNIT; }
    } else {
      // Helper mode: tile 1 of conv_layer5 is the helper's, and conv_layer5 consumes its units in the order 1, 0, 2, 3, 4, 5
      // (C5U) in the bands and in the helper alike -- the helper starts a block with the channels it computed itself.  (The
      // kernel without helpers keeps 0, 1, ..: the other order costs it 2.5 %, a scheduling accident of hipcc; the two
      // forms therefore differ in the last bits of conv_layer5's sums.)  During conv_layer5 W.wp stays on its first block.
      const float* wsrc = W.wp;
      if (K == 3 && u == U - 1) wsrc = W.wp + 2 * C5U(0) * UNIT;
      else if (K == 4 && u + 1 < U) wsrc = W.wp + 2 * C5U(u + 1) * UNIT;
      else if (K == 4) { wsrc = W.wp + 12 * UNIT; W.wp += 13 * UNIT; }
      else W.wp += UNIT;
      issue_loads<1>(nxt, wsrc, lane);
    }
    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the prefetch to just before its first use
    const int b = W.bofs + (w + 8 * ((HM && K == 4) ? C5U(u) : u)) * 4 * CS;
    const bool newest = (K == 0) || (u == U - 1);
    if (HM && K == 0 && u == 1) {
      if (from_helper) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) ok = ok && (W.hh_l[r] < 0 || (unsigned)(hv[r] >> 32) == tag_in);
        if (!__all(ok)) {  // the helper is behind: poll (the three requests of a round are in flight together)
          int spins = 0;
#pragma nounroll
          do {
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int r = 0; r < 3; ++r) hv[r] = granule_load(hp[r]);
            ok = true;
#pragma unroll
            for (int r = 0; r < 3; ++r) ok = ok && (W.hh_l[r] < 0 || (unsigned)(hv[r] >> 32) == tag_in);
            if (++spins > SPIN_LIMIT) { *a.err = 1; *a.err_dev = 1; break; }
          } while (!__all(ok));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (W.hh_l[r] >= 0) lds[W.hh_l[r] + (w + 8) * 4 * CS] = __uint_as_float((unsigned)hv[r]);
      }
      mma_taps<TP, NM, 1>(cur, b, acc);  // (the tap order of a newest unit, as in the kernel without helpers)
      asm volatile("" : "+v"(acc[0]));
      mma_taps<TP, NM, 2>(cur, b, acc);
    } else if (!newest) {
      mma_taps<TP, NM, 0>(cur, b, acc);
    } else {
      mma_taps<TP, NM, 1>(cur, b, acc);
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
      mma_taps<TP, NM, 2>(cur, b, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (LATE && u == 0) {
      layer_requests();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#ifndef TF_NO_EARLY_WAIT
  {
    // The NEXT layer's first weights (requested a unit ago) are awaited HERE, before the epilogue's global and granule stores:
    // vmcnt counts stores as well, the stores sit in divergent branches, and behind a branch hipcc can only wait with vmcnt(0) --
    // the wait at the next layer's start then covered the full write latency of this layer's stores, 1 900 cycles per layer
    // (TF_TIMING: 11 % of the launch).  Here nothing younger than the weights is in flight yet.
    constexpr bool lastpar = ((base + U - 1) & 1) != 0;
    float (&pend)[36] = lastpar ? A0 : A1;
#pragma unroll
    for (int i = 0; i < 36; ++i) asm volatile("" ::"v"(pend[i]));
    __builtin_amdgcn_sched_barrier(0);
    if (TF_LATE_LOADS) {
      // ... and the NEXT layer's second unit is requested here, into the buffer the last unit has just finished with
      // (what that layer's first unit used to request at its start, right behind the stores below)
      float (&freebuf)[36] = lastpar ? A1 : A0;
      issue_second_unit<(K + 1) % 5, HM>(freebuf, W, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#endif
  TF_LAP(0);
  // ---- split-K reduction over the eight wavefronts + epilogue, one 32-channel tile at a time ----
  const unsigned tag_out = ((unsigned)a.epoch << 12) | (unsigned)(serial + 1);
  const int par_out = serial & 1;
  // uniform bases of this layer's outputs
  float* gbase = nullptr;
  if (K < 4) { if (a.store_all) gbase = a.cat[j] + (64 + 32 * K) * 81; }
  else { gbase = a.store_all ? a.cat[j + 1] : (last_rdb ? a.out : nullptr); }
  unsigned long long* obox = a.inbox + (par_out ? 2 * 64 * HS : 0);
  const bool publish = !(K == 4 && last_rdb);
  unsigned long long* hbox = HM ? a.inbox + a.off_hb + (((size_t)W.cl * 2 + (j & 1)) * 5 + K) * (32 * 81) : nullptr;
  constexpr int LCH = K < 4 ? (64 + 32 * K) * CS : 0;  // first LDS plane of this layer's output
#pragma unroll
  for (int mt = 0; mt < NM; ++mt) {
    if (mt) __syncthreads();  // the previous tile's sums have been read
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[RED0 + (w * 16 + r) * 64 + lane] = acc[mt][r];
    TF_LAP(2);
    __syncthreads();
    TF_LAP(1);
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
        if (W.up_ok) {  // my first positions are what follows the previous tile
#pragma unroll
          for (int i = 0; i < 2; ++i) granule_store(obox + (W.ep_up + (mt * 32 + i) * HS), v[i], tag_out, W.local);
        }
        if (W.dn_ok) {  // my last positions are what precedes the next tile
#pragma unroll
          for (int i = 0; i < 2; ++i) granule_store(obox + (W.ep_dn + (mt * 32 + i) * HS), v[i], tag_out, W.local);
        }
        if (HM) {
#pragma unroll
          for (int i = 0; i < 2; ++i) granule_store(hbox + (W.ep_h + i * 81), v[i], tag_out, W.local);
        }
      }
    }
  }
  TF_LAP(2);
  __syncthreads();  // planes written: the next layer may read them
  TF_LAP(3);
}


// ---- helper mode: the fourth workgroup of an image ----
// Holds the WHOLE image's dense-block concat in LDS (192 planes of 11 x 10 cells, zero-framed) and computes output
// channels 32..63 of conv_layer5 for the three bands: the same split-K over eight wavefronts, every A operand used for three
// B tiles.  Channel quad q belongs to wavefront q % 8 here as well; a wavefront takes its quads' values from the bands'
// granules (324 consecutive granules per quad: 4 channels x 81 positions), one unit ahead of the MFMAs that need them.
// Unit order per dense block: 1 (its own channels 32..63 of the block input), 0 (the bands' channels 0..31), 2 .. 5 (the
// bands' conv_layer1..4 outputs, as they appear); taps and the eight-way sum as in the bands.
constexpr int CSH = 112;
constexpr int REDH = 192 * CSH;
constexpr size_t LDS_HELPER = (size_t)(REDH + NWAVE * 16 * 64) * sizeof(float);

DI void helper_issue(float (&A)[18], const float* p, int lane) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f4v v = *reinterpret_cast<const f4v*>(p + c * 256 + lane * 4);
    A[4 * c + 0] = v.x; A[4 * c + 1] = v.y; A[4 * c + 2] = v.z; A[4 * c + 3] = v.w;
  }
  const f2v u = *reinterpret_cast<const f2v*>(p + 1024 + lane * 2);
  A[16] = u.x; A[17] = u.y;
}

// SEL as in mma_taps (TP = 27): 0 all taps, 1 the middle kernel row, 2 the outer rows
template <int SEL> DI void helper_taps(const float (&A)[18], int b0, f16v (&acc)[3]) {
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (SEL == 1 && tap / 3 != 1) continue;
    if (SEL == 2 && tap / 3 == 1) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float bv = lds[b0 + 30 * b + ks * 2 * CSH + (tap / 3) * 10 + tap % 3];
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[tap * 2 + ks], bv, acc[b], 0, 0, 0);
      }
    }
  }
}

DI void helper_trunk(const Args& a, const int cl) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int n = lane & 31;
  const bool st_ok = n < 27;
  const int img = a.img0 + cl;
  const int nq = st_ok ? n : 0;
  const int cell0 = (nq / 9 + 1) * 10 + nq % 9 + 1;   // band 0; band b: + 30 b
  const int bofs = (lane >> 5) * CSH + cell0 - 11;
  const int m0 = ((2 * w) & 3) + 8 * ((2 * w) >> 2) + 4 * (lane >> 5);
  const unsigned xtag = ((unsigned)a.epoch << 12) | 0xFFFu;   // (layer serials stay below 0xFFF)
  const bool local = a.local_st && xcd_handshake(a.inbox + a.off_xcc, 4 * cl, 3, 0x7u, xtag, lane);
  for (int i = t; i < 192 * CSH; i += NTHREADS) lds[i] = 0.f;
  __syncthreads();
  const float* in = a.in + (size_t)img * 192 * 81;
  for (int i = t; i < 64 * 81; i += NTHREADS) {
    const int ch = i / 81, q = i % 81;
    lds[ch * CSH + (q / 9 + 1) * 10 + q % 9 + 1] = in[ch * 81 + q];
  }
  float xres[3][2];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int i = 0; i < 2; ++i) xres[b][i] = st_ok ? in[(32 + m0 + i) * 81 + 27 * b + n] : 0.f;
  __syncthreads();
  int fcell[6];   // granule s = lane + 64 r of a quad's 4 x 81 block -> LDS cell relative to the quad's first plane
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int s0 = lane + 64 * r;
    const int e = s0 < 324 ? s0 / 81 : 0, q = s0 < 324 ? s0 % 81 : 0;
    fcell[r] = s0 < 324 ? e * CSH + (q / 9 + 1) * 10 + q % 9 + 1 : -1;
  }
  const float* wbase = a.wstream + (size_t)w * a.nrdb * WAVE_RDB + 15 * UNIT;   // conv_layer5, unit 0, tile 1; unit u: + 2 u UNIT
  const unsigned long long* hb = a.inbox + a.off_hb + (size_t)cl * (2 * 5 * 32 * 81) + 4 * w * 81 + lane;
  unsigned long long* bh = a.inbox + a.off_bh + (size_t)cl * (2 * 32 * 81);
  float A0[18], A1[18];
  helper_issue(A0, wbase + 2 * C5U(0) * UNIT, lane);
  for (int j = 0; j < a.nrdb; ++j) {
    const bool last_rdb = j == a.nrdb - 1;
    const float* wj = wbase + (size_t)j * WAVE_RDB;
    f16v acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    float bias[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bias[i] = a.bstream[j * 192 + 160 + m0 + i];
    unsigned long long gv[6];
    const unsigned long long* gsrc = nullptr;
    unsigned gtag = 0;
#pragma unroll
    for (int step = 0; step < 6; ++step) {
      const int u = C5U(step);   // 1 (this workgroup's own channels 32..63 of the block input), 0, 2, 3, 4, 5
      float (&cur)[18] = (step & 1) ? A1 : A0;
      float (&nxt)[18] = (step & 1) ? A0 : A1;
#pragma unroll
      for (int i = 0; i < 18; ++i) asm volatile("" ::"v"(cur[i]));
      __builtin_amdgcn_sched_barrier(0);
      // the next step's weights (the next block's first step after the last one; the stream is padded at its end)
      helper_issue(nxt, step < 5 ? wj + 2 * C5U(step + 1) * UNIT : wj + WAVE_RDB + 2 * C5U(0) * UNIT, lane);
      __builtin_amdgcn_sched_barrier(0);
      // ---- this unit's inputs: requested one step ago (gv), validated and written to this wavefront's planes now ----
      const bool fetched = !(u == 1 || (u == 0 && j == 0));
      if (fetched) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 6; ++r) ok = ok && (fcell[r] < 0 || (unsigned)(gv[r] >> 32) == gtag);
        if (!__all(ok)) {  // the bands are behind: poll (the six requests of a round are in flight together)
          int spins = 0;
#pragma nounroll
          do {
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int r = 0; r < 6; ++r) gv[r] = granule_load(fcell[r] >= 0 ? gsrc + 64 * r : gsrc);
            ok = true;
#pragma unroll
            for (int r = 0; r < 6; ++r) ok = ok && (fcell[r] < 0 || (unsigned)(gv[r] >> 32) == gtag);
            if (++spins > SPIN_LIMIT) { *a.err = 1; *a.err_dev = 1; break; }
          } while (!__all(ok));
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if (fcell[r] >= 0) lds[fcell[r] + (w + 8 * u) * 4 * CSH] = __uint_as_float((unsigned)gv[r]);
      }
      // ---- request the next step's inputs ----
      if (step < 5) {
        const int un = C5U(step + 1);
        if (un == 0) {  // the bands' channels 0..31 of the previous block's output
          gsrc = hb + (size_t)((((j + 1) & 1) * 5 + 4) * (32 * 81));
          gtag = ((unsigned)a.epoch << 12) | (unsigned)(j * 5);
        } else {        // conv_layer(un - 1) of this block
          gsrc = hb + (size_t)(((j & 1) * 5 + (un - 2)) * (32 * 81));
          gtag = ((unsigned)a.epoch << 12) | (unsigned)(j * 5 + un - 1);
        }
        if (!(un == 0 && j == 0)) {
#pragma unroll
          for (int r = 0; r < 6; ++r) gv[r] = granule_load(fcell[r] >= 0 ? gsrc + 64 * r : gsrc);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- 18 x 3 MFMAs; the last unit in the tap order of a band's newest unit (middle kernel row first) ----
      const int b0 = bofs + (w + 8 * u) * 4 * CSH;
      if (u == 5) {
        helper_taps<1>(cur, b0, acc);
        helper_taps<2>(cur, b0, acc);
      } else {
        helper_taps<0>(cur, b0, acc);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- split-K reduction + epilogue, band by band ----
    const unsigned tag_out = ((unsigned)a.epoch << 12) | (unsigned)(j * 5 + 5);
    float* gbase = a.store_all ? a.cat[j + 1] : (last_rdb ? a.out : nullptr);
    unsigned long long* obox = bh + (size_t)((j & 1) * (32 * 81));
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (b) __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[REDH + (w * 16 + r) * 64 + lane] = acc[b][r];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 2 * w + i;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < NWAVE; ++ww) v += lds[REDH + (ww * 16 + r) * 64 + lane];
        v += bias[i];
        const int lc = (32 + m0 + i) * CSH + cell0 + 30 * b;
        v = a.rs * v + lds[lc];                      // a5 * rs + a0  (:358)
        if (j % 3 == 2) {                            // a3 * rs + x   (:402)
          v = a.rs * v + xres[b][i];
          xres[b][i] = v;
        }
        if (st_ok) {
          lds[lc] = v;
          if (gbase) gbase[((size_t)img * 192 + 32 + m0 + i) * 81 + 27 * b + n] = v;
          if (!last_rdb) granule_store(obox + (m0 + i) * 81 + 27 * b + n, v, tag_out, local);
        }
      }
    }
    __syncthreads();
  }
}

template <int TP, bool HM>
__global__ __launch_bounds__(512) void trunk_fused_kernel(Args a) {
  constexpr int CS = Geo<TP>::CS;
  Wave W;
  W.t = threadIdx.x; W.lane = W.t & 63; W.w = W.t >> 6;
  // block b runs on XCD b % 8 (observed; speed only): an XCD owns a contiguous run of tiles, so neighbours share its L2
  // (helper mode: the three bands and the helper of an image are four consecutive blocks of one XCD)
  const int B = blockIdx.x;
  int tile;
  if (HM) {
    const int cl = ((B / 8) / 4) * 8 + B % 8, role = (B / 8) % 4;
    if (cl >= a.nimg) return;
    if (role == 3) { helper_trunk(a, cl); return; }
    tile = cl * 3 + role;
    W.cl = cl;
  } else {
    tile = (B % 8) * a.tpx + B / 8;
    W.cl = tile / 3;   // (TP = 27: a.tpx is a multiple of three -- an image's bands are consecutive blocks of one XCD)
  }
  if (tile >= a.ntiles) return;
  W.local = false;
  if (TP == 27 && a.local_st) {  // partners: the bands above / below (and the image's helper)
    const int role = tile - 3 * W.cl;
    const unsigned mask = (role > 0 ? 1u << (role - 1) : 0u) | (role < 2 ? 1u << (role + 1) : 0u) | (HM ? 8u : 0u);
    W.local = xcd_handshake(a.inbox + a.off_xcc, 4 * W.cl, role, mask, ((unsigned)a.epoch << 12) | 0xFFFu, W.lane);
  }
  // the tile's own positions [P0, pend) of the launch's images laid end to end; (i0, r0) / (il, rl): image and row of the
  // first / last one.  LDS row of (image i, row r) = (i - i0) * 10 + r - r0 + 1: row 0 is the row above the first own row,
  // and between the last row of image i0 and the first row of image i0 + 1 lies one row nobody writes (zero).
  const int total = a.nimg * 81;
  const int P0 = tile * TP;
  const int pend = P0 + TP < total ? P0 + TP : total;
  const int i0 = P0 / 81, r0 = (P0 % 81) / 9;
  const int il = (pend - 1) / 81, rl = ((pend - 1) % 81) / 9;
  auto cell = [&](int p) { const int i = p / 81, q = p % 81; return ((i - i0) * 10 + q / 9 - r0 + 1) * 10 + q % 9 + 1; };
  // position hp outside [P0, pend): does a tap of an own position reach it?
  auto halo_ok = [&](int hp) {
    if (hp < P0) return hp >= 0 && hp / 81 == i0 && (hp % 81) / 9 >= r0 - 1;
    return tile + 1 < a.ntiles && hp < total && hp / 81 == il && (hp % 81) / 9 <= rl + 1;
  };
  W.n = W.lane & 31;
  W.st_ok = W.n < TP && P0 + W.n < pend;
  const int pc = W.st_ok ? P0 + W.n : P0;   // (padding lanes compute on the first own cell; their results go nowhere)
  const int img = pc / 81, q = pc % 81;
  W.pofs = cell(pc);
  W.bofs = (W.lane >> 5) * CS + W.pofs - 11;
  W.wp = a.wstream + (size_t)W.w * a.nrdb * WAVE_RDB;
  {
    const int m0 = ((2 * W.w) & 3) + 8 * ((2 * W.w) >> 2) + 4 * (W.lane >> 5), n = W.n;
    const unsigned me = (unsigned)tile;
    // my first HS positions follow the previous tile, my last HS precede the next one (same image only)
    W.up_ok = W.st_ok && n < HS && tile > 0 && img == (P0 - 1) / 81;
    W.dn_ok = W.st_ok && n >= TP - HS && tile + 1 < a.ntiles && img == (P0 + TP) / 81;
    W.ep_l = (unsigned)(m0 * CS + W.pofs);
    W.ep_g = (unsigned)(((a.img0 + img) * 192 + m0) * 81 + q);
    W.ep_up = (((me - 1) * 2) * 2 + 1) * (64 * HS) + m0 * HS + n;               // [tile][parity][side][64][HS]
    W.ep_dn = (((me + 1) * 2) * 2 + 0) * (64 * HS) + m0 * HS + (n - (TP - HS));
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int s0 = W.lane + 64 * r;
      const int sl = s0 < 8 * HS ? s0 : 0;
      const int e = sl / (2 * HS), side = (sl / HS) & 1, c = sl % HS;
      const int hp = side == 0 ? P0 - HS + c : P0 + TP + c;
      const bool have = s0 < 8 * HS && halo_ok(hp);
      W.hl_g[r] = ((me * 2) * 2 + side) * (64 * HS) + e * HS + c;
      W.hl_l[r] = have ? e * CS + cell(hp) : -1;
    }
    W.ep_h = (unsigned)(m0 * 81 + q);
#pragma unroll
    for (int r = 0; r < 3; ++r) {  // (helper mode: TP = 27, the tile is three whole rows of image `img`)
      const int s0 = W.lane + 64 * r;
      const int e = s0 < 180 ? s0 / 45 : 0, pp = s0 < 180 ? s0 % 45 : 0;
      const int qq = P0 % 81 - 9 + pp;  // from the row above the band's first row
      const bool have = HM && s0 < 180 && qq >= 0 && qq < 81;
      W.hh_g[r] = (unsigned)(e * 81 + (have ? qq : 0));
      W.hh_l[r] = have ? e * CS + cell(img * 81 + qq) : -1;
    }
  }

  for (int i = W.t; i < 192 * CS; i += NTHREADS) lds[i] = 0.f;
  __syncthreads();
  // trunk input: channels 0..63 at the own positions and the halo positions either side
  for (int i = W.t; i < 64 * (TP + 2 * HS); i += NTHREADS) {
    const int ch = i / (TP + 2 * HS), k = i % (TP + 2 * HS);
    const int hp = P0 - HS + k;
    const bool ok = (k >= HS && k < HS + TP) ? hp < pend : halo_ok(hp);
    if (ok) lds[ch * CS + cell(hp)] = a.in[((size_t)(a.img0 + hp / 81) * 192 + ch) * 81 + hp % 81];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // this thread's conv5 outputs: tile mt = k / 2, register r = 2 w + (k & 1)
    const int r = 2 * W.w + (k & 1);
    const int m = (k >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (W.lane >> 5);
    W.xres[k] = W.st_ok ? a.in[((size_t)(a.img0 + img) * 192 + m) * 81 + q] : 0.f;
  }
  __syncthreads();

#ifdef TF_TIMING
  for (int i = 0; i < 8; ++i) W.tsum[i] = 0;
  const long long t_begin = TF_NOW();
#endif
#ifdef TF_PRIO
  // the second wavefront of every SIMD (dispatched later = the loser of every issue arbitration by age) gets a static priority
  if (W.w >= NWAVE / 2) __builtin_amdgcn_s_setprio(TF_PRIO);
#endif
  float A0[36], A1[36];
  issue_loads<1>(A0, W.wp, W.lane);
  W.wp += UNIT;
  if (TF_LATE_LOADS) issue_second_unit<0, HM>(A1, W, W.lane);
  for (int j = 0; j < a.nrdb; ++j) {
    const bool last = j == a.nrdb - 1;
    dense_layer<TP, 0, HM>(a, W, A0, A1, j, last);
    dense_layer<TP, 1, HM>(a, W, A0, A1, j, last);
    dense_layer<TP, 2, HM>(a, W, A0, A1, j, last);
    dense_layer<TP, 3, HM>(a, W, A0, A1, j, last);
    dense_layer<TP, 4, HM>(a, W, A0, A1, j, last);
  }
#ifdef TF_TIMING
  W.tsum[7] = TF_NOW() - t_begin;
  if (W.lane == 0 && a.tstamp)
    for (int i = 0; i < 8; ++i) a.tstamp[((size_t)blockIdx.x * NWAVE + W.w) * 8 + i] = W.tsum[i];
#endif
}

// The trunk's weights into the per-wavefront streams (one launch per optimizer step).  One wavefront per (unit, tile) block:
// a lane's eighteen values are two runs of nine consecutive floats of the OIHW tensor (its output channel, input channels
// 2 ks + (lane >> 5) of the quad, all taps), written as the four 16-byte + one 8-byte pieces issue_loads reads back -- no
// LDS, no per-element index arithmetic (the element-wise gather this replaces: 47 us, 1.4 TB/s).
__global__ __launch_bounds__(256) void pack_trunk_fused_kernel(const float* const* wsrc, const float* const* bsrc, float* wstream,
                                                               float* bstream, int nrdb) {
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);       // (wavefront w, dense block j, block bidx)
  const int ntask = nrdb * NWAVE * 26;
  if (task < ntask) {
    const int w = task / (nrdb * 26), j = (task / 26) % nrdb, bidx = task % 26;
    int K, u, mt = 0;
    if (bidx < 2) { K = 0; u = bidx; }
    else if (bidx < 5) { K = 1; u = bidx - 2; }
    else if (bidx < 9) { K = 2; u = bidx - 5; }
    else if (bidx < 14) { K = 3; u = bidx - 9; }
    else { K = 4; u = (bidx - 14) / 2; mt = (bidx - 14) % 2; }
    const int Cin = 64 + 32 * K;
    const int cout = mt * 32 + (lane & 31);
    const float* src = wsrc[j * 5 + K] + ((long)cout * Cin + 4 * (w + 8 * u) + (lane >> 5)) * 9;
    float A[18];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) A[tap * 2 + ks] = src[ks * 18 + tap];
    float* dst = wstream + ((size_t)w * nrdb + j) * WAVE_RDB + (size_t)bidx * UNIT;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<f4v*>(dst + c * 256 + lane * 4) = (f4v){A[4 * c], A[4 * c + 1], A[4 * c + 2], A[4 * c + 3]};
    *reinterpret_cast<f2v*>(dst + 1024 + lane * 2) = (f2v){A[16], A[17]};
  }
  const long nb = (long)nrdb * 192;
  for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nb; f += (long)gridDim.x * blockDim.x) {
    const int j = (int)(f / 192), c = (int)(f % 192);
    const int K = c < 128 ? c / 32 : 4;
    bstream[f] = bsrc[j * 5 + K][c - 32 * K];
  }
}

// ------------------------------------------------------------------------------------------------------------------
size_t trunk_fused_stream_floats(int nrdb) { return (size_t)nrdb * NWAVE * WAVE_RDB + 8 * UNIT; }  // + read-ahead pad
// the neighbour boxes [tiles][2][2][64][HS] at three tiles per image, the helpers' inboxes and the boxes they fill (the
// backward chain's boxes are smaller)
size_t trunk_fused_inbox_bytes(int nimg) {
  return ((size_t)3 * nimg * 2 * 2 * 64 * HS + (size_t)nimg * 2 * 5 * 32 * 81 + (size_t)nimg * 2 * 32 * 81 + (size_t)4 * nimg) * sizeof(unsigned long long);
}
// granules in front of the XCC_ID table [image][4] at the end of an inbox buffer sized for `nimg_alloc` images
size_t trunk_fused_xcc_offset(int nimg_alloc) { return trunk_fused_inbox_bytes(nimg_alloc) / sizeof(unsigned long long) - (size_t)4 * nimg_alloc; }
bool g_trunk_local_off = false;  // set by the first time-out event of the process: from then on agent-scope exchange stores only
int trunk_local_stores() {
  static const int v = DBM_TUNE_GETENV("TRUNK_LOCAL_ST") ? atoi(DBM_TUNE_GETENV("TRUNK_LOCAL_ST")) : 1;
  return v && !g_trunk_local_off;
}

void launch_pack_trunk_fused(const float* const* d_wsrc, const float* const* d_bsrc, float* wstream, float* bstream, int nrdb,
                             hipStream_t s) {
  hipLaunchKernelGGL(pack_trunk_fused_kernel, dim3((nrdb * NWAVE * 26 + 3) / 4), dim3(256), 0, s, d_wsrc, d_bsrc, wstream, bstream, nrdb);
  DBM_HIP(hipGetLastError());
}

void launch_trunk_fused(const TrunkFusedLaunch& L, hipStream_t s) {
  // DBM_TRUNK_TP=27 (default): three whole rows per tile (three workgroups per image, 27 of 32 MFMA columns); 32: tiles of 32
  // consecutive positions across rows and images (5184 positions of 64 images = 162 workgroups, every column works; measured
  // 5 % slower per pass: only the centre tap can run ahead of the halo, 255 VGPRs).
  // DBM_TRUNK_HELPER (TP = 27): a fourth workgroup per image takes half of conv_layer5 (4 x 64 = all 256 CUs; 1.29 -> 1.09 ms
  // per 64-image pass).  Bit 0: in passes that keep nothing (inference, the D-step's fakes), bit 1: in retained passes.
  // Default 1: the G-step's retained forward runs underneath the discriminator's passes, which need the 64 CUs it leaves
  // (measured: 8.82 ms per step without helpers, 8.75 with bit 0, 9.2 with both).
  static const int TP = getenv("DBM_TRUNK_TP") ? atoi(getenv("DBM_TRUNK_TP")) : 27;
  static const int helper_mask = getenv("DBM_TRUNK_HELPER") ? atoi(getenv("DBM_TRUNK_HELPER")) : 1;
  static const int n_cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return prop.multiProcessorCount;
  }();
  // (four workgroups per image, all resident at once, one per CU: only if the device has that many)
  const bool helper = TP == 27 && (helper_mask & (L.cat ? 2 : 1)) != 0 && 4 * L.nimg <= n_cus && !L.no_helper;
  DBM_CHECK(TP == 27 || TP == 32, "DBM_TRUNK_TP must be 27 or 32");
  static bool attr = false;
  if (!attr) {
    DBM_HIP(hipFuncSetAttribute((const void*)trunk_fused_kernel<27, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Geo<27>::LDS_BYTES));
    DBM_HIP(hipFuncSetAttribute((const void*)trunk_fused_kernel<27, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_HELPER));
    DBM_HIP(hipFuncSetAttribute((const void*)trunk_fused_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)Geo<32>::LDS_BYTES));
    attr = true;
  }
  Args a;
  a.wstream = L.wstream; a.bstream = L.bstream; a.in = L.in; a.out = L.out;
  a.store_all = L.cat != nullptr;
  for (int i = 0; i < TRUNK_FUSED_MAXCAT; ++i) a.cat[i] = (L.cat && i <= L.nrdb) ? L.cat[i] : nullptr;
  DBM_CHECK(L.nrdb + 1 <= TRUNK_FUSED_MAXCAT, "fused trunk: too many dense blocks");
  a.inbox = L.inbox; a.err = L.err; a.err_dev = L.err_dev;
  a.nrdb = L.nrdb; a.nimg = L.nimg; a.img0 = L.img0; a.epoch = L.epoch & 0xFFFFF; a.rs = L.rs; a.slope = L.slope;
  a.ntiles = (L.nimg * 81 + TP - 1) / TP;
  a.tpx = TP == 27 ? 3 * ((L.nimg + 7) / 8) : (a.ntiles + 7) / 8;
  a.off_xcc = (unsigned)trunk_fused_xcc_offset(64);
  a.local_st = trunk_local_stores();
  a.off_hb = (unsigned)(3 * L.nimg * 2 * 2 * 64 * HS);
  a.off_bh = a.off_hb + (unsigned)(L.nimg * 2 * 5 * 32 * 81);
  const double flop = 2.0 * 19408896.0 * L.nrdb * L.nimg;  // 19 408 896 MAC per dense block and tile
  if (g_profiler.enabled) {
    // algorithmic bytes: the 26 624 x 9 weights of every dense block once, the trunk's input once, and what the launch has to
    // leave in HBM -- the 64-channel output, plus (training: store_all) the 192-channel concatenation of every dense block
    const double wbytes = 4.0 * 26624.0 * 9.0 * L.nrdb;
    const double abytes = 4.0 * 81.0 * L.nimg * (64.0 + 64.0 + (L.cat ? 192.0 * L.nrdb : 0.0));
    char tag[40];
    snprintf(tag, sizeof(tag), "trunk_fwd_%drdb_n%d%s%s", L.nrdb, L.nimg, L.cat ? "_keep" : "", helper ? "_helper" : "");
    g_profiler.begin(s, helper ? 4 : 2, flop, wbytes + abytes, tag, helper ? 32 * ((L.nimg + 7) / 8) : 8 * a.tpx);
  }
#ifdef TF_TIMING
  static long long* d_ts = nullptr;
  if (!d_ts) DBM_HIP(hipMalloc((void**)&d_ts, sizeof(long long) * 512 * NWAVE * 8));
  DBM_HIP(hipMemsetAsync(d_ts, 0, sizeof(long long) * 512 * NWAVE * 8, s));
  a.tstamp = d_ts;
#endif
  if (helper)
    hipLaunchKernelGGL((trunk_fused_kernel<27, true>), dim3(32 * ((L.nimg + 7) / 8)), dim3(NTHREADS), LDS_HELPER, s, a);
  else if (TP == 27)
    hipLaunchKernelGGL((trunk_fused_kernel<27, false>), dim3(8 * a.tpx), dim3(NTHREADS), Geo<27>::LDS_BYTES, s, a);
  else
    hipLaunchKernelGGL((trunk_fused_kernel<32, false>), dim3(8 * a.tpx), dim3(NTHREADS), Geo<32>::LDS_BYTES, s, a);
  if (g_profiler.enabled) g_profiler.end(s);
#ifdef TF_TIMING
  if (g_profiler.enabled && g_profiler.serial) {   // (the serialised profile pass of tools/experiments/step_shapes.py)
    std::vector<long long> h((size_t)512 * NWAVE * 8);
    DBM_HIP(hipDeviceSynchronize());
    DBM_HIP(hipMemcpy(h.data(), d_ts, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int b : {0, 8, 16}) {   // (helper form: blocks 0, 8, 16 are the three bands of image 0; block 24 its helper)
      for (int w = 0; w < NWAVE; ++w) {
        const long long* t = &h[((size_t)b * NWAVE + w) * 8];
        fprintf(stderr, "trunk fwd timing (%s) block %3d wave %d: K loops %9lld  barrier1 %8lld  reduce+epilogue %8lld  barrier2 %8lld  prologue: before the first load %8lld, loads %8lld  total %9lld\n",
                helper ? "helper" : "retained", b, w, t[0], t[1], t[2], t[3], t[5], t[4], t[7]);
      }
    }
  }
#endif
  DBM_HIP(hipGetLastError());
}
