// Micro-benchmark: do a wavefront's own vector instructions run under its own MFMAs?  One wavefront per SIMD (or two), a loop body of
// NM MFMAs (alternating two accumulators) and NV independent packed-fp32 FMAs dealt between them (inline asm: exact placement), timed
// with s_memtime.  Question behind it (round 6): the LDS-window deformable kernels' step loops take about the SUM of their vector
// arithmetic and their MFMAs, not the maximum.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// KIND 0: v_mfma_f32_32x32x2f32 (64 cycles), 1: v_mfma_f32_32x32x16_bf16 (32 cycles).  VPER: packed FMAs behind every MFMA.
// VK: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_cvt_pk_bf16_f32, 3 v_and_b32
template <int KIND, int VPER, bool MF, int VK = 0>
__global__ __launch_bounds__(512) void k(long long* out, int iters, float* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f16v acc[2];
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  f2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = (f2){1e-3f * (lane + i), 2e-3f * (lane - i)};
  const f2 m = {1.0001f, 0.9999f};
  float a = 1e-3f * lane, b = 2e-3f * lane;
  bf8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(1e-2f * (lane + i)); bb[i] = (__bf16)(1e-2f * (lane - i)); }
  __syncthreads();
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (MF) {
        if (KIND == 0) acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 1], 0, 0, 0);
        else acc[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[q & 1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < VPER; ++j) {
        f2& x = v[(q * VPER + j) & 7];
        if (VK == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(m));
        else if (VK == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[0]) : "v"(m[0]));
        else if (VK == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[0]) : "v"(x[1]));
        else asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[0]));
      }
      asm volatile("" : "+v"(acc[0]), "+v"(acc[1]) : : );   // (keeps the MFMAs where they are written; costs nothing)
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}

template <int KIND, int VPER, bool MF, int VK = 0>
static void run(const char* what, int waves) {
  long long* out; float* sink;
  hipMalloc((void**)&out, 64 * 8 * sizeof(long long)); hipMalloc((void**)&sink, 4);
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, VPER, MF, VK>), dim3(64), dim3(64 * waves), 0, 0, out, iters, sink);
  hipLaunchKernelGGL((k<KIND, VPER, MF, VK>), dim3(64), dim3(64 * waves), 0, 0, out, iters, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(64 * 8);
  hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  // (the slowest wavefront of the workgroup: with two per SIMD the older one wins every arbitration and finishes early)
  double sum = 0;
  for (int b = 0; b < 64; ++b) {
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[b * 8 + w] > mx ? h[b * 8 + w] : mx;
    sum += (double)mx;
  }
  printf("%-46s %d wavefront(s)/SIMD: %8.2f cycles per body of 8 MFMA slots (slowest wavefront)\n", what, waves / 4, sum / 64 / iters);
  hipFree(out); hipFree(sink);
}

int main() {
  for (int waves : {4, 8}) {
    run<0, 0, true>("fp32 32x32x2: MFMAs only", waves);
    run<0, 6, false>("fp32: 6 packed FMAs per slot, no MFMA", waves);
    run<0, 6, true>("fp32: MFMA + 6 packed FMAs per slot", waves);
    run<0, 12, false>("fp32: 12 packed FMAs per slot, no MFMA", waves);
    run<0, 12, true>("fp32: MFMA + 12 packed FMAs per slot", waves);
    run<1, 0, true>("bf16 32x32x16: MFMAs only", waves);
    run<1, 4, false>("bf16: 4 packed FMAs per slot, no MFMA", waves);
    run<1, 4, true>("bf16: MFMA + 4 packed FMAs per slot", waves);
    run<1, 8, false>("bf16: 8 packed FMAs per slot, no MFMA", waves);
    run<1, 8, true>("bf16: MFMA + 8 packed FMAs per slot", waves);
    run<1, 8, false, 1>("bf16: 8 v_fma_f32 per slot, no MFMA", waves);
    run<1, 8, true, 1>("bf16: MFMA + 8 v_fma_f32 per slot", waves);
    run<1, 8, false, 2>("bf16: 8 v_cvt_pk_bf16_f32 per slot, no MFMA", waves);
    run<1, 8, true, 2>("bf16: MFMA + 8 v_cvt_pk_bf16_f32 per slot", waves);
    run<1, 8, false, 3>("bf16: 8 v_and_b32 per slot, no MFMA", waves);
    run<1, 8, true, 3>("bf16: MFMA + 8 v_and_b32 per slot", waves);
    run<0, 12, false, 1>("fp32: 12 v_fma_f32 per slot, no MFMA", waves);
    run<0, 12, true, 1>("fp32: MFMA + 12 v_fma_f32 per slot", waves);
  }
  return 0;
}
