// Micro-benchmark: what does ONE wavefront (or two per SIMD) achieve in fp32 MFMAs per cycle when every MFMA takes its B operand
// from a fresh ds_read_b32, as the persistent trunk kernels do?   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
extern __shared__ float lds[];

// MODE 0: 16x16x4, registers only; 1: 16x16x4, B from LDS (one ds_read_b32 per MFMA, prefetched 9 ahead); 2: 32x32x2 registers; 3: 32x32x2 LDS
template <int MODE, int NACC>
__global__ __launch_bounds__(512) void k(long long* out, int iters, float* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  float a[9];
  for (int i = 0; i < 9; ++i) a[i] = 1e-3f * (lane + i);
  const int b0 = (lane >> 4) * 80 + (lane & 15) + w * 640;
  long long t0 = 0;
  if constexpr (MODE < 2) {
    f4v acc[NACC];
    for (int c = 0; c < NACC; ++c) acc[c] = (f4v){0, 0, 0, 0};
    float bq[2][9];
    for (int t = 0; t < 9; ++t) bq[0][t] = lds[b0 + (t / 3) * 10 + t % 3];
    __syncthreads();
    t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (MODE == 1) {
#pragma unroll
          for (int t = 0; t < 9; ++t) bq[(q + 1) & 1][t] = lds[b0 + ((q + 1) & 3) * 320 + (t / 3) * 10 + t % 3];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[(q * 9 + t) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], MODE == 1 ? bq[q & 1][t] : a[(t + 1) % 9], acc[(q * 9 + t) % NACC], 0, 0, 0);
        if (MODE == 1) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  } else {
    f16v acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
    const int b1 = (lane >> 5) * 80 + (lane & 31) + w * 640;
    __syncthreads();
    t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float bv = MODE == 3 ? lds[b1 + q * 160 + (t / 3) * 40 + t % 3] : a[(t + 1) % 9];
          acc[(q * 9 + t) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[(q * 9 + t) % NACC], 0, 0, 0);
        }
      }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < NACC; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  }
}

template <int MODE, int NACC> void run(const char* name, int waves, long long* d_out, float* d_sink) {
  const int iters = 2000, blocks = 256;
  const int per_iter = MODE < 2 ? 36 : 18;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(64 * waves), 65536, 0, d_out, iters, d_sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(64 * waves), 65536, 0, d_out, iters, d_sink);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double tf = 2.0 * (MODE < 2 ? 1024.0 : 2048.0) * per_iter * iters * waves * blocks / (ms * 1e-3) / 1e12;
  std::vector<long long> h(blocks * 8);
  hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) s += (double)h[b * 8 + w];
  const double cyc = s / (blocks * waves) / (iters * per_iter);
  const int ideal = MODE < 2 ? 32 : 64;
  printf("%-34s waves/WG %d (%.0f per SIMD): %.1f ticks per MFMA per wave -> SIMD pipe busy %.0f %%; wall %.3f ms = %.1f TFLOP/s\n", name, waves, waves / 4.0, cyc,
         100.0 * ideal * (waves / 4.0) / cyc, ms, tf);
}

int main() {
  long long* d_out; float* d_sink;
  hipMalloc(&d_out, 256 * 8 * sizeof(long long)); hipMalloc(&d_sink, 16);
  hipFuncSetAttribute((const void*)k<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int waves : {4, 8}) {
    run<0, 1>("16x16x4 regs NACC=1", waves, d_out, d_sink);
    run<0, 2>("16x16x4 regs NACC=2", waves, d_out, d_sink);
    run<0, 4>("16x16x4 regs NACC=4", waves, d_out, d_sink);
    run<1, 2>("16x16x4 B from LDS NACC=2", waves, d_out, d_sink);
    run<1, 4>("16x16x4 B from LDS NACC=4", waves, d_out, d_sink);
    run<2, 1>("32x32x2 regs NACC=1", waves, d_out, d_sink);
    run<2, 2>("32x32x2 regs NACC=2", waves, d_out, d_sink);
    run<3, 1>("32x32x2 B from LDS NACC=1", waves, d_out, d_sink);
    run<3, 2>("32x32x2 B from LDS NACC=2", waves, d_out, d_sink);
  }
  return 0;
}
