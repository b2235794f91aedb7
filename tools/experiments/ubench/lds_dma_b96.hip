// Where does `global_load_lds` put a lane's bytes?  (VERDICT r5 #8b: conv_tile.hip's 12-byte staging form "gave wrong results".)
// One wavefront; lane l loads `SIZE` bytes from src + l * SIZE (words tagged 1000 * l + k) into LDS at M0-base + lane-linear offset;
// the LDS is then dumped and the landing dword index of every (lane, k) is printed for SIZE = 4, 12, 16.
//   hipcc --offload-arch=gfx950 -O2 -o lds_dma_b96 lds_dma_b96.hip && ./lds_dma_b96
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
template <int SIZE> __global__ void probe(const float* src, float* out) {
  __shared__ __attribute__((aligned(16))) float buf[512];
  for (int i = threadIdx.x; i < 512; i += 64) buf[i] = -1.f;
  __syncthreads();
  const char* p = (const char*)src + threadIdx.x * SIZE;
#if defined(__HIP_DEVICE_COMPILE__)   // (the size must be a literal)
  if constexpr (SIZE == 4) __builtin_amdgcn_global_load_lds((const void*)p, (lds_ptr)buf, 4, 0, 0);
  if constexpr (SIZE == 12) __builtin_amdgcn_global_load_lds((const void*)p, (lds_ptr)buf, 12, 0, 0);
  if constexpr (SIZE == 16) __builtin_amdgcn_global_load_lds((const void*)p, (lds_ptr)buf, 16, 0, 0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}
template <int SIZE> void run(const float* d_src, float* d_out) {
  hipLaunchKernelGGL(probe<SIZE>, dim3(1), dim3(64), 0, 0, d_src, d_out);
  std::vector<float> h(512);
  hipMemcpy(h.data(), d_out, 512 * 4, hipMemcpyDeviceToHost);
  printf("SIZE %2d bytes per lane: LDS dword -> (lane, word) for the first 20 dwords:", SIZE);
  for (int i = 0; i < 20; ++i) {
    if (h[i] < 0) printf(" [%d: untouched]", i);
    else printf(" [%d: l%d.w%d]", i, (int)h[i] / 1000, (int)h[i] % 1000);
  }
  int lane_stride = -1, bad = 0, touched = 0;
  for (int i = 0; i < 512; ++i) {
    if (h[i] < 0) continue;
    ++touched;
    const int l = (int)h[i] / 1000, k = (int)h[i] % 1000;
    if (l == 1 && k == 0) lane_stride = i;
    if (i != l * (SIZE / 4) + k) ++bad;
  }
  printf("\n   dwords written %d (expected %d); lane 1's first word landed at dword %d (lane-linear would be %d); %d dwords off the lane-linear "
         "layout dst + lane * SIZE\n", touched, 64 * SIZE / 4, lane_stride, SIZE / 4, bad);
}
int main() {
  std::vector<float> src(64 * 4);
  for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) src[l * 4 + k] = 0.f;
  float *d_src, *d_out;
  hipMalloc(&d_src, 4096); hipMalloc(&d_out, 4096);
  // source words are tagged by the (lane, word) that SHOULD fetch them under a lane-linear reading of the source: word w of the buffer
  // belongs to lane w / (SIZE / 4), word index w % (SIZE / 4) -- rebuilt per SIZE
  for (int size : {4, 12, 16}) {
    std::vector<float> s(1024, 0.f);
    for (int w = 0; w < 64 * size / 4; ++w) s[w] = 1000.f * (w / (size / 4)) + (w % (size / 4));
    hipMemcpy(d_src, s.data(), 4096, hipMemcpyHostToDevice);
    if (size == 4) run<4>(d_src, d_out);
    if (size == 12) run<12>(d_src, d_out);
    if (size == 16) run<16>(d_src, d_out);
  }
  return 0;
}
