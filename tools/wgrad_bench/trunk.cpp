// standalone micro-benchmark of the batched weight-gradient kernel on a synthetic trunk group
#include "wgrad_abl.hip"
#include <cstdlib>
KernelProfiler g_profiler;
void KernelProfiler::begin(hipStream_t, int, double, double, const char*, long) {}
void KernelProfiler::end(hipStream_t) {}
void KernelProfiler::collect(double*, int) {}
int main(int argc, char** argv) {
  const int N = 64, h = 9, w = 9, hw = 81;
  const int nrdb = argc > 1 ? atoi(argv[1]) : 12;
  float *x, *dy, *gw;
  const size_t bufsz = (size_t)N * 192 * hw;
  hipMalloc(&x, bufsz * 4 * nrdb); hipMalloc(&dy, bufsz * 4 * nrdb); hipMalloc(&gw, 4 * 400000 * (size_t)nrdb);
  std::vector<float> hx(bufsz * nrdb);
  for (auto& v : hx) v = (rand() % 1000) * 1e-3f;
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemset(gw, 0, 4 * 400000 * (size_t)nrdb);
  WgradBatch b;
  double fl = 0;
  for (int j = 0; j < nrdb; ++j) {
    float* g = gw + (size_t)j * 400000;
    for (int k = 0; k < 5; ++k) {
      WgradDesc d; memset(&d, 0, sizeof(d));
      const int cin = 64 + 32 * k, cout = k == 4 ? 64 : 32;
      d.x = x + bufsz * j; d.xsn = 192 * hw; d.xsc = hw; d.Cin = cin; d.Hin = h; d.Win = w;
      d.dy = dy + bufsz * j + (size_t)(k == 4 ? 0 : cin) * hw; d.dysn = 192 * hw; d.dysc = hw; d.Cout = cout; d.OH = h; d.OW = w;
      d.KH = d.KW = 3; d.stride = 1; d.pad = 1; d.N = N; d.scale = 1.f; d.gW = g; d.gb = gw + (size_t)j * 400000 + 399000;
      g += (size_t)cout * cin * 9;
      b.add(d);
      fl += 2.0 * N * hw * cout * cin * 9;
    }
  }
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) b.launch(s);
  hipStreamSynchronize(s);
  const int reps = 20;
  hipEventRecord(e0, s);
  for (int i = 0; i < reps; ++i) b.launch(s);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  {
    static unsigned long long h[4 * 8192];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof(h));
    const int nwg = b.total_wg[3] ? b.total_wg[3] : 1;
    double a[4] = {0, 0, 0, 0};
    for (int i = 0; i < nwg && i < 8192; ++i) for (int k = 0; k < 4; ++k) a[k] += (double)h[4 * i + k];
    printf("per task cycles: stage %.0f kloop %.0f epilogue %.0f total %.0f (n=%d)\n", a[0] / nwg, a[1] / nwg, a[2] / nwg, a[3] / nwg, nwg);
  }
  printf("%d RDB: %d workgroups, %.1f us per launch, %.1f TFLOP/s\n", nrdb, b.total_wg[1] + b.total_wg[3] + b.total_wg[4], 1e3 * ms / reps, fl / (ms / reps * 1e-3) / 1e12);
  return 0;
}
