// Throughput of device-scope (agent) atomics from many workgroups: one address for all against one address per workgroup.
// Why: round 6's one-launch deep end of the discriminator (ticket-scheduled work items, round6_calls/19_*.patch) pays two such atomics
// per work item -- one ticket draw on ONE counter, one arrival on its stage's counter -- and ran at ~120 ns per item.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_same_address atomic_same_address.hip && ./atomic_same_address
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SCOPE>
__global__ __launch_bounds__(256) void draw(unsigned* ctr, int stride, int iters, unsigned* sink) {
  unsigned v = 0;
  if (threadIdx.x == 0) {
    unsigned* p = ctr + (size_t)blockIdx.x * stride;
    for (int i = 0; i < iters; ++i) v += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, SCOPE);   // (the result is used: a returning atomic, like a ticket)
    sink[blockIdx.x] = v;
  }
}

int main() {
  unsigned *ctr, *sink;
  CK(hipMalloc((void**)&ctr, 1024 * 64 * sizeof(unsigned)));
  CK(hipMalloc((void**)&sink, 1024 * sizeof(unsigned)));
  CK(hipMemset(ctr, 0, 1024 * 64 * sizeof(unsigned)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 64;
  for (int wgs : {8, 64, 256, 512}) {
    for (int stride : {0, 64}) {   // 0: one counter for all workgroups; 64 words = 256 B: a line of its own per workgroup
      float ms[2] = {0, 0};
      for (int scope = 0; scope < 2; ++scope) {
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(a));
          if (scope == 0) hipLaunchKernelGGL(draw<__HIP_MEMORY_SCOPE_AGENT>, dim3(wgs), dim3(256), 0, 0, ctr, stride, iters, sink);
          else hipLaunchKernelGGL(draw<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(wgs), dim3(256), 0, 0, ctr, stride, iters, sink);
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          CK(hipEventElapsedTime(&ms[scope], a, b));
        }
      }
      printf("%3d workgroups x %d returning atomics, %s: agent scope %7.1f us = %6.1f ns per atomic (whole chip), %6.2f us per workgroup-serial atomic; "
             "workgroup scope %7.1f us = %6.1f ns\n", wgs, iters, stride ? "own line each" : "ONE address  ", 1e3 * ms[0], 1e6 * ms[0] / (wgs * iters),
             1e3 * ms[0] / iters, 1e3 * ms[1], 1e6 * ms[1] / (wgs * iters));
    }
  }
  return 0;
}
