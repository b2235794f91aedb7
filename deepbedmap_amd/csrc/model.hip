// Shared model machinery: tensor tables, arenas, packed-weight caches, conv descriptors.
#include "model.h"

static const size_t GUARD = 32;  // floats on each side: the row-tap loads of igemm touch one word outside a tensor

void DevBuf::ensure(size_t count, bool zero) {
  if (count > n) {
    if (p) DBM_HIP(hipFree(p - GUARD));
    p = nullptr;
    float* base = nullptr;
    DBM_HIP(hipMalloc((void**)&base, (count + 2 * GUARD) * sizeof(float)));
    n = count;
    zero = true;  // fresh allocations are always zeroed (padding channels rely on it)
    DBM_HIP(hipMemset(base, 0, (count + 2 * GUARD) * sizeof(float)));
    p = base + GUARD;
    // the memset runs on the NULL stream, which the context's non-blocking stream does not wait for
    DBM_HIP(hipDeviceSynchronize());
    return;
  }
  (void)zero;
}

void dbm_ctx::fork_to_side(int k) {
  DBM_HIP(hipEventRecord(ev_fork[k & 7], stream));
  DBM_HIP(hipStreamWaitEvent(side, ev_fork[k & 7], 0));
}

void dbm_ctx::persist_begin(hipStream_t s) {
  if (persist_pending) DBM_HIP(hipStreamWaitEvent(s, ev_persist, 0));
}
void dbm_ctx::persist_end(hipStream_t s) {
  if (!ev_persist) DBM_HIP(hipEventCreateWithFlags(&ev_persist, hipEventDisableTiming));
  DBM_HIP(hipEventRecord(ev_persist, s));
  persist_pending = true;
}

void dbm_ctx::fork(hipStream_t from, hipStream_t to, int k) {
  DBM_HIP(hipEventRecord(ev_fork[k & 7], from));
  DBM_HIP(hipStreamWaitEvent(to, ev_fork[k & 7], 0));
}

void dbm_ctx::join_side() {
  DBM_HIP(hipEventRecord(ev_join, side));
  DBM_HIP(hipStreamWaitEvent(stream, ev_join, 0));
}

void DevBuf::release() {
  if (p) (void)hipFree(p - GUARD);
  p = nullptr;
  n = 0;
}

dbm_model::~dbm_model() {
  if (ctx)
    for (size_t i = 0; i < ctx->models.size(); ++i)
      if (ctx->models[i] == this) { ctx->models.erase(ctx->models.begin() + i); break; }
  if (is_view) return;  // nothing here is owned
  if (d_adam_skipped) (void)hipFree(d_adam_skipped);
  for (auto& L : layers) {
    if (L.wf) (void)hipFree(L.wf);
    if (L.wf16) (void)hipFree(L.wf16);
    if (L.wcl16) (void)hipFree(L.wcl16);
    if (L.wx3) (void)hipFree(L.wx3);
    if (L.wdx3) (void)hipFree(L.wdx3);
    for (int i = 0; i < 4; ++i)
      if (L.wb[i]) (void)hipFree(L.wb[i]);
  }
  if (params) (void)hipFree(params);
  if (grads) (void)hipFree(grads);
  if (adam_m) (void)hipFree(adam_m);
  if (adam_v) (void)hipFree(adam_v);
  if (pers) (void)hipFree(pers);
  if (d_pack_jobs) (void)hipFree(d_pack_jobs);
  if (d_bwd_jobs) (void)hipFree(d_bwd_jobs);
  if (d_lazy_jobs) (void)hipFree(d_lazy_jobs);
}

void dbm_model::mark_grads_touched() {
  for (dbm_model* o : ctx->models)
    if (o->grads == grads) o->grads_touched = true;
  grads_touched = true;
}

int dbm_model::add_tensor(const std::string& key, std::vector<int64_t> shape, int kind) {
  Tensor t;
  t.key = key;
  t.ndim = (int)shape.size();
  t.n = 1;
  for (int i = 0; i < 4; ++i) t.shape[i] = i < t.ndim ? shape[i] : 1;
  for (int i = 0; i < t.ndim; ++i) t.n *= (size_t)shape[i];
  t.kind = kind;
  size_t& cursor = kind == DBM_KIND_PARAM ? nparam : npers;
  t.off = cursor;
  cursor += t.n;
  tensors.push_back(t);
  index[key] = (int)tensors.size() - 1;
  return (int)tensors.size() - 1;
}

void dbm_model::alloc_arenas() {
  const size_t np = nparam ? nparam : 1, ns = npers ? npers : 1;
  DBM_HIP(hipMalloc((void**)&params, np * sizeof(float)));
  DBM_HIP(hipMalloc((void**)&grads, np * sizeof(float)));
  DBM_HIP(hipMalloc((void**)&adam_m, np * sizeof(float)));
  DBM_HIP(hipMalloc((void**)&adam_v, np * sizeof(float)));
  DBM_HIP(hipMalloc((void**)&pers, ns * sizeof(float)));
  DBM_HIP(hipMemset(params, 0, np * sizeof(float)));
  DBM_HIP(hipMemset(grads, 0, np * sizeof(float)));
  DBM_HIP(hipMemset(adam_m, 0, np * sizeof(float)));
  DBM_HIP(hipMemset(adam_v, 0, np * sizeof(float)));
  DBM_HIP(hipMemset(pers, 0, ns * sizeof(float)));
  DBM_HIP(hipMalloc((void**)&d_adam_skipped, 256));
  DBM_HIP(hipMemset(d_adam_skipped, 0, 256));
  if (ctx) ctx->models.push_back(this);
  DBM_HIP(hipDeviceSynchronize());  // NULL-stream memsets vs. the context's non-blocking stream
}

int dbm_model::tid(const std::string& key) const {
  auto it = index.find(key);
  DBM_CHECK(it != index.end(), "unknown tensor key " + key);
  return it->second;
}

static inline int up32(int v) { return (v + 31) & ~31; }

// Registers a convolution whose tensors `name/W` (+ `name/b`) were already added.
int dbm_model::add_iglayer(const std::string& name, int O, int C, int K, int stride, int pad, bool bias, bool as_1x1) {
  IgLayer L;
  L.wi = tid(name + "/W");
  L.bi = bias ? tid(name + "/b") : -1;
  L.O = O; L.C = C; L.K = K; L.stride = stride; L.pad = pad;
  L.Cview = as_1x1 ? C * K * K : C;
  L.Kview = as_1x1 ? 1 : K;
  L.CinP = up32(L.Cview);
  L.CoutP = up32(O);
  L.OP = up32(O);
  L.CP = up32(L.Cview);
  const int T = L.Kview * L.Kview;
  DBM_CHECK(T <= DBM_MAX_TAPS, "kernel too large for the igemm path");
  DBM_HIP(hipMalloc((void**)&L.wf, sizeof(float) * (size_t)T * L.CinP * L.CoutP));
  if (stride == 1) {
    L.Tb = T;
    for (int t = 0; t < T; ++t) {
      const int ky = t / L.Kview, kx = t % L.Kview;
      L.bky[0][t] = (signed char)ky; L.bkx[0][t] = (signed char)kx;
      L.bdy[0][t] = (signed char)(L.pad - ky); L.bdx[0][t] = (signed char)(L.pad - kx);
    }
    DBM_HIP(hipMalloc((void**)&L.wb[0], sizeof(float) * (size_t)T * L.OP * L.CP));
    // a K x K layer viewed as 1x1 over (c, tap) columns (the deformable convolution's GEMM): also its per-tap transposed
    // image [tap][o][c], the A operand of the fused column-gradient kernel (deform_bwd64_fused_kernel)
    if (as_1x1 && K > 1 && K * K <= DBM_MAX_TAPS) DBM_HIP(hipMalloc((void**)&L.wb[1], sizeof(float) * (size_t)K * K * up32(O) * up32(C)));
  } else {
    DBM_CHECK(stride == 2 && L.Kview == 4 && pad == 1, "strided igemm layers must be k4 s2 p1");
    L.Tb = 4;
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
      int t = 0;
      for (int ky = 0; ky < 4; ++ky) {
        if (((py + 1 - ky) & 1) != 0) continue;
        for (int kx = 0; kx < 4; ++kx) {
          if (((px + 1 - kx) & 1) != 0) continue;
          L.bky[ph][t] = (signed char)ky; L.bkx[ph][t] = (signed char)kx;
          L.bdy[ph][t] = (signed char)((py + 1 - ky) / 2); L.bdx[ph][t] = (signed char)((px + 1 - kx) / 2);
          ++t;
        }
      }
      DBM_HIP(hipMalloc((void**)&L.wb[ph], sizeof(float) * (size_t)4 * L.OP * L.CP));
    }
  }
  layers.push_back(L);
  return (int)layers.size() - 1;
}

// job table of the one-launch repack for the layers with lazy == want_lazy
// which: 1 = the forward images, 2 = the data-gradient images, 3 = both
static void build_pack_table(const dbm_model& m, bool want_lazy, int which, PackJob** d_jobs, int* njobs, int* nblocks) {
  std::vector<PackJob> jobs;
  int blocks = 0;
  auto add = [&](const IgLayer& L, int T, const signed char* ky, const signed char* kx, int transpose, int KP, int MP,
                 float* dst) {
    PackJob j;
    memset(&j, 0, sizeof(j));
    j.w = m.P(L.wi); j.dst = dst; j.O = L.O; j.C = L.Cview; j.KH = L.Kview; j.KW = L.Kview; j.T = T;
    j.transpose = transpose; j.KP = KP; j.MP = MP;
    for (int t = 0; t < T; ++t) { j.ky[t] = ky[t]; j.kx[t] = kx[t]; }
    // one workgroup per 32 x TC tile of the (out, in) plane (pack_weights_kernel)
    const int taps = L.Kview * L.Kview, TC = taps > 9 ? 16 : 32;
    DBM_CHECK(taps <= 16 && KP % 32 == 0 && MP % 32 == 0, "pack_weights_kernel: unsupported layer geometry");
    const int oR = transpose ? KP : MP, cR = transpose ? MP : KP;
    const int nb = (oR / 32) * (cR / TC);
    j.block_start = blocks; j.block_count = nb;
    blocks += nb;
    jobs.push_back(j);
  };
  for (auto& L : m.layers) {
    if (L.lazy != want_lazy) continue;
    const int T = L.Kview * L.Kview;
    signed char ky[DBM_MAX_TAPS], kx[DBM_MAX_TAPS];
    for (int t = 0; t < T; ++t) { ky[t] = (signed char)(t / L.Kview); kx[t] = (signed char)(t % L.Kview); }
    if (which & 1) add(L, T, ky, kx, 0, L.CinP, L.CoutP, L.wf);
    if (!(which & 2)) continue;
    const int nph = L.stride == 1 ? 1 : 4;
    for (int ph = 0; ph < nph; ++ph) add(L, L.Tb, L.bky[ph], L.bkx[ph], 1, L.OP, L.CP, L.wb[ph]);
    if (L.stride == 1 && L.wb[1]) {  // per-tap transposed image of a 1x1-viewed K x K layer: dst[t][o][c] = W[o][c][ky][kx]
      PackJob j;
      memset(&j, 0, sizeof(j));
      const int KK = L.K * L.K;
      DBM_CHECK(KK <= DBM_MAX_TAPS, "pack: kernel too large");
      j.w = m.P(L.wi); j.dst = L.wb[1]; j.O = L.O; j.C = L.C; j.KH = L.K; j.KW = L.K; j.T = KK;
      j.transpose = 1; j.KP = up32(L.O); j.MP = up32(L.C);
      for (int t = 0; t < KK; ++t) { j.ky[t] = (signed char)(t / L.K); j.kx[t] = (signed char)(t % L.K); }
      const int TC = KK > 9 ? 16 : 32;
      const int nb = (j.KP / 32) * (j.MP / TC);
      j.block_start = blocks; j.block_count = nb;
      blocks += nb;
      jobs.push_back(j);
    }
  }
  *njobs = (int)jobs.size();
  *nblocks = blocks;
  if (!jobs.empty()) {
    DBM_HIP(hipMalloc((void**)d_jobs, jobs.size() * sizeof(PackJob)));
    DBM_HIP(hipMemcpy(*d_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
    DBM_HIP(hipDeviceSynchronize());
  }
}

void dbm_model::ensure_packed(hipStream_t on) {
  if (!packed_dirty) return;
  static const int abl_nopack = DBM_MEASURE_ENV("ABL_NOPACK");  // (libdbm_measure.so only; results wrong)
  if (pack_tables_built && type == 1 && (abl_nopack & 1)) { packed_dirty = false; bwd_dirty = false; return; }
  if (pack_tables_built && type == 0 && (abl_nopack & 6)) {
    hipStream_t s2 = on ? on : ctx->stream;
    if (!(abl_nopack & 2) && n_pack_jobs) launch_pack_jobs(d_pack_jobs, n_pack_jobs, n_pack_blocks, s2);
    if (!(abl_nopack & 4)) pack_extra(s2);
    packed_dirty = false;
    bwd_dirty = !(abl_nopack & 2);
    return;
  }
  hipStream_t s = on ? on : ctx->stream;
  if (!pack_tables_built) {  // the job tables only depend on the layer list: build and upload them once
    build_pack_table(*this, false, 1, &d_pack_jobs, &n_pack_jobs, &n_pack_blocks);
    build_pack_table(*this, false, 2, &d_bwd_jobs, &n_bwd_jobs, &n_bwd_blocks);
    build_pack_table(*this, true, 3, &d_lazy_jobs, &n_lazy_jobs, &n_lazy_blocks);
    pack_tables_built = true;
  }
  if (n_pack_jobs) launch_pack_jobs(d_pack_jobs, n_pack_jobs, n_pack_blocks, s);
  pack_extra(s);
  packed_dirty = false;
  bwd_dirty = true;
  static const bool split = !(getenv("DBM_PACK_SPLIT") && atoi(getenv("DBM_PACK_SPLIT")) == 0);
  if (!split) ensure_packed_bwd(s);
}

void dbm_model::ensure_packed_bwd(hipStream_t on) {
  if (is_view) return;
  if (packed_dirty) ensure_packed(on);
  if (!bwd_dirty) return;
  if (n_bwd_jobs) launch_pack_jobs(d_bwd_jobs, n_bwd_jobs, n_bwd_blocks, on ? on : ctx->stream);
  bwd_dirty = false;
}

// Layers marked lazy (the generator's trunk when the persistent kernels serve it) keep their per-layer images only for
// the callers that still need them: other plane sizes than 9x9, DBM_TRUNK_FUSED=0.
void dbm_model::ensure_packed_lazy(hipStream_t on) {
  ensure_packed(on);
  if (lazy_version == param_version || !n_lazy_jobs) return;
  launch_pack_jobs(d_lazy_jobs, n_lazy_jobs, n_lazy_blocks, on ? on : ctx->stream);
  lazy_version = param_version;
}

// bf16 forward images (DBM_BF16 inference): dst[t][g][kh][co][i] = bf16(W[co][cin = 16 g + 8 kh + i][tap t])
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int O, int Cview,
                                                        int Kview, int CinP, int CoutP) {
  const int T = Kview * Kview, G = CinP / 16;
  const long total = (long)T * G * 2 * CoutP * 8;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int i = (int)(e & 7);
    long r = e >> 3;
    const int co = (int)(r % CoutP); r /= CoutP;
    const int kh = (int)(r & 1); r >>= 1;
    const int g = (int)(r % G);
    const int t = (int)(r / G);
    const int cin = 16 * g + 8 * kh + i;
    float v = 0.f;
    if (co < O && cin < Cview) v = w[((long)co * Cview + cin) * T + t];  // OIHW, taps fastest
    dst[e] = (__bf16)v;
  }
}

void dbm_model::ensure_packed_bf16() {
  if (packed16_version == param_version) return;
  hipStream_t s = ctx->stream;
  for (auto& L : layers) {
    const int T = L.Kview * L.Kview;
    const size_t n = (size_t)T * L.CinP * L.CoutP;
    if (!L.wf16) DBM_HIP(hipMalloc(&L.wf16, n * sizeof(__bf16)));
    long nb = ((long)n + 2047) / 2048;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, s, P(L.wi), (__bf16*)L.wf16, L.O, L.Cview, L.Kview, L.CinP,
                       L.CoutP);
    if (L.want_cl16 && L.K == 3 && L.C % 32 == 0 && (L.O == 32 || L.O == 64)) {  // fragment-ordered image (conv_cl16.hip)
      if (!L.wcl16) DBM_HIP(hipMalloc(&L.wcl16, cl16_packed_elems(L.C, L.O) * sizeof(__bf16)));
      launch_pack_cl16(P(L.wi), L.wcl16, L.O, L.C, s);
    }
    if (L.want_x3 && L.K == 3 && L.Kview == 3 && L.C % 16 == 0 && L.O <= 64) {  // split-bf16 image (conv_cl16x3_kernel)
      if (!L.wx3) DBM_HIP(hipMalloc(&L.wx3, cl16x3_packed_elems(L.C, L.O) * sizeof(__bf16)));
      launch_pack_cl16x3(P(L.wi), L.wx3, L.O, L.C, s);
    }
    if (L.want_dx3 && L.K == 3 && L.C == 64 && L.O == 64) {
      if (!L.wdx3) DBM_HIP(hipMalloc(&L.wdx3, deform_x3_packed_elems() * sizeof(__bf16)));
      launch_pack_deform_x3(P(L.wi), L.wdx3, s);
    }
  }
  DBM_HIP(hipGetLastError());
  packed16_version = param_version;
}

ConvDesc dbm_model::fwd_desc(const IgLayer& L, const float* x, long xsn, int Hin, int Win, int ups, float* y, long ysn,
                             int N) const {
  ConvDesc d;
  memset(&d, 0, sizeof(d));
  const int Hl = Hin << ups, Wl = Win << ups;
  const int OH = (Hl + 2 * L.pad - L.Kview) / L.stride + 1, OW = (Wl + 2 * L.pad - L.Kview) / L.stride + 1;
  d.x = x; d.xsn = xsn; d.xsc = Hin * Win; d.Cin = L.CinP; d.Hin = Hin; d.Win = Win; d.ups = ups;
  d.N = N; d.OHl = OH; d.OWl = OW; d.sin = L.stride;
  d.T = L.Kview * L.Kview;
  for (int t = 0; t < d.T; ++t) {
    d.dy[t] = (signed char)(t / L.Kview - L.pad);
    d.dx[t] = (signed char)(t % L.Kview - L.pad);
  }
  d.wp = L.wf; d.CoutP = L.CoutP; d.Cout = L.O;
  d.wp16 = use_bf16 ? L.wf16 : nullptr;
  d.bias = L.bi >= 0 ? P(L.bi) : nullptr;
  d.y = y; d.ysn = ysn; d.ysc = OH * OW; d.OWp = OW; d.so = 1;
  d.s1 = 1.f; d.r1s = 1.f; d.s2 = 1.f; d.slope = 0.2f;
  d.zeros = ctx->zeros;
  return d;
}

// Data gradient of layer L.  `base` carries the gradient input (x, xsn = dY and its image stride), the output
// (y, ysn) and every epilogue field; geometry and weights are filled here.  Hin_fwd/Win_fwd: forward INPUT dims
// (after upsample), i.e. the dims of the gradient being produced.
void dbm_model::run_dgrad(const IgLayer& L, ConvDesc base, int Hin_fwd, int Win_fwd, hipStream_t s) const {
  if (!s) s = ctx->stream;
  DBM_CHECK(is_view || !bwd_dirty, "run_dgrad: the data-gradient weight images are stale (ensure_packed_bwd was not called)");
  const int OH = (Hin_fwd + 2 * L.pad - L.Kview) / L.stride + 1, OW = (Win_fwd + 2 * L.pad - L.Kview) / L.stride + 1;
  base.xsc = OH * OW; base.Cin = L.OP; base.Hin = OH; base.Win = OW; base.ups = 0;
  static const int cin_live_env = getenv("DBM_CIN_LIVE") ? atoi(getenv("DBM_CIN_LIVE")) : 1;   // (A/B switch)
  base.cin_live = (cin_live_env && L.O < L.OP) ? ((L.O + 7) & ~7) : 0;   // (gradient channels past the layer's O outputs are zero padding)
  base.sin = 1;
  base.CoutP = L.CP; base.Cout = L.Cview;
  base.bias = nullptr;
  base.ysc = Hin_fwd * Win_fwd; base.OWp = Win_fwd;
  base.zeros = ctx->zeros;
  base.slope = 0.2f;
  if (L.stride == 1) {
    base.OHl = Hin_fwd; base.OWl = Win_fwd; base.so = 1; base.oy0 = 0; base.ox0 = 0;
    base.T = L.Tb;
    for (int t = 0; t < L.Tb; ++t) { base.dy[t] = L.bdy[0][t]; base.dx[t] = L.bdx[0][t]; }
    base.wp = L.wb[0];
    launch_igemm_conv(base, s);
  } else {
    // the four phases (py, px) of the stride-2 gradient -- output positions (2a + py, 2b + px), 2x2 taps each -- as ONE launch
    // (blockIdx.z = phase); planes that lack a phase (a single row or column) fall back to one launch per phase
    static const int merge = DBM_TUNE_GETENV("IGEMM_MERGE_PHASES") ? atoi(DBM_TUNE_GETENV("IGEMM_MERGE_PHASES")) : 1;
    base.so = 2; base.T = 4;
    if (merge && Hin_fwd >= 2 && Win_fwd >= 2) {
      base.nphase = 4;
      for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        base.phOH[ph] = (short)((Hin_fwd - py + 1) / 2); base.phOW[ph] = (short)((Win_fwd - px + 1) / 2);
        for (int t = 0; t < 4; ++t) { base.dy[4 * ph + t] = L.bdy[ph][t]; base.dx[4 * ph + t] = L.bdx[ph][t]; }
        base.phwp[ph] = L.wb[ph];
      }
      base.OHl = base.phOH[0]; base.OWl = base.phOW[0]; base.oy0 = 0; base.ox0 = 0; base.wp = L.wb[0];
      launch_igemm_conv(base, s);
      return;
    }
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
      base.OHl = (Hin_fwd - py + 1) / 2; base.OWl = (Win_fwd - px + 1) / 2;
      if (base.OHl <= 0 || base.OWl <= 0) continue;
      base.oy0 = py; base.ox0 = px;
      for (int t = 0; t < 4; ++t) { base.dy[t] = L.bdy[ph][t]; base.dx[t] = L.bdx[ph][t]; }
      base.wp = L.wb[ph];
      launch_igemm_conv(base, s);
    }
  }
}

void dbm_model::run_wgrad(const IgLayer& L, const float* x, long xsn, int Hin, int Win, int ups, const float* dy,
                          long dysn, int OH, int OW, int N, float scale, WgradBatch* batch) const {
  WgradDesc w;
  memset(&w, 0, sizeof(w));
  w.x = x; w.xsn = xsn; w.xsc = Hin * Win; w.Cin = L.Cview; w.Hin = Hin; w.Win = Win; w.ups = ups;
  w.dy = dy; w.dysn = dysn; w.dysc = OH * OW; w.Cout = L.O; w.OH = OH; w.OW = OW;
  w.KH = L.Kview; w.KW = L.Kview; w.stride = L.stride; w.pad = L.Kview == 1 ? 0 : L.pad;
  w.N = N; w.scale = scale;
  w.gW = G(L.wi);
  w.gb = L.bi >= 0 ? G(L.bi) : nullptr;
  if (batch) batch->add(w);
  else launch_wgrad(w, ctx->stream);
}
