// per-layer micro-benchmark of the discriminator's weight gradients (N = 64, 36x36 input)
#include "wgrad_abl.hip"
#include <cstdlib>
KernelProfiler g_profiler;
void KernelProfiler::begin(hipStream_t, int, double, double, const char*, long) {}
void KernelProfiler::end(hipStream_t) {}
void KernelProfiler::collect(double*, int) {}
int main(int argc, char** argv) {
  const int N = 64;
  static const int O[10] = {64, 64, 128, 128, 128, 256, 256, 512, 512, 512};
  static const int Ci[10] = {1, 64, 64, 128, 128, 128, 256, 256, 512, 512};
  static const int K[10] = {3, 4, 3, 4, 3, 4, 3, 4, 3, 4};
  static const int S[10] = {1, 2, 1, 2, 1, 2, 1, 2, 1, 2};
  // NOTE the reference's channel plan: conv4 is 128->128? use the table of discriminator.hip
  int hs[11]; hs[0] = 36; hs[1] = 36;
  for (int i = 1; i < 10; ++i) hs[i + 1] = (hs[i] + 2 - K[i]) / S[i] + 1;
  const size_t big = (size_t)N * 64 * 36 * 36;
  float *x, *dy, *gw;
  hipMalloc(&x, big * 4); hipMalloc(&dy, big * 4); hipMalloc(&gw, 4 * 512 * 512 * 16 + 4096);
  std::vector<float> hx(big);
  for (auto& v : hx) v = (rand() % 1000) * 1e-3f;
  hipMemcpy(x, hx.data(), big * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, hx.data(), big * 4, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double tot = 0;
  for (int i = 1; i < 10; ++i) {
    WgradDesc d; memset(&d, 0, sizeof(d));
    const int hin = hs[i], ho = hs[i + 1];
    d.x = x; d.xsn = (long)Ci[i] * hin * hin; d.xsc = hin * hin; d.Cin = Ci[i]; d.Hin = hin; d.Win = hin;
    d.dy = dy; d.dysn = (long)O[i] * ho * ho; d.dysc = ho * ho; d.Cout = O[i]; d.OH = ho; d.OW = ho;
    d.KH = d.KW = K[i]; d.stride = S[i]; d.pad = 1; d.N = N; d.scale = 1.f; d.gW = gw; d.gb = nullptr;
    WgradBatch b; b.add(d);
    for (int r = 0; r < 2; ++r) b.launch(s);
    hipStreamSynchronize(s);
    const int reps = 10;
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) b.launch(s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * N * ho * ho * O[i] * Ci[i] * K[i] * K[i];
    int wgs = 0, cat = -1; for (int c = 0; c < WgradBatch::NCAT; ++c) if (b.total_wg[c]) { wgs = b.total_wg[c]; cat = c; }
    printf("conv%d %3d->%3d k%d s%d %2dx%2d -> %2dx%2d: cat %d, %5d wgs, %7.1f us, %5.1f TFLOP/s (%.2f GF)\n", i, Ci[i], O[i], K[i], S[i], hin, hin, ho, ho, cat, wgs,
           1e3 * ms / reps, fl / (ms / reps * 1e-3) / 1e12, fl * 1e-9);
    tot += 1e3 * ms / reps;
    {
      static unsigned long long h[4 * 8192];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof(h));
      const int nwg = wgs < 8192 ? wgs : 8192;
      double a[4] = {0, 0, 0, 0};
      for (int q = 0; q < nwg; ++q) for (int k = 0; k < 4; ++k) a[k] += (double)h[4 * q + k];
      if (cat >= 3) printf("      per group cycles: stage %.0f kloop %.0f epilogue %.0f total %.0f\n", a[0] / nwg, a[1] / nwg, a[2] / nwg, a[3] / nwg);
    }
  }
  printf("sum %.1f us per slot\n", tot);
  return 0;
}
