// BatchNorm, dense head, losses and Adam: the HBM-bound (non-GEMM) part of the training step.
#include "dbm_internal.h"
#include "kernels.h"

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* sh) {  // sh: NT / 64 floats
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += sh[w];
  return t;
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ----------------------------------------------------------------------------------------------
// L.BatchNormalization(axis=(0,2,3), eps=1e-5) + F.leaky_relu   (srgan_train.py:636-644, 664-689)
// One workgroup per channel: two-pass mean / biased variance (as Chainer's x.mean / x.var), running
// statistics with the unbiased correction m/max(m-1,1) and decay 0.9, then normalise + LeakyReLU.
// ----------------------------------------------------------------------------------------------
// NT = 256: flat index over (n, p) (tiny planes: the deep layers, 512 channels = 512 workgroups).  NT = 1024: one
// wavefront per image, lanes along the plane (the 64..128-channel layers on 18x18 / 9x9 planes would otherwise run on
// 64 workgroups of 256 threads: 66 us for a 21 MB tensor).
template <int NT>
__global__ __launch_bounds__(NT) void bn_train_fwd_kernel(const float* __restrict__ z, float* __restrict__ y,
                                                          const float* gamma, const float* beta, float* mean_o,
                                                          float* istd_o, float* avg_mean, float* avg_var, int N, int C,
                                                          int plane, float eps, float decay, float slope, const int* hold) {
  __shared__ float sh[NT / 64];
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  constexpr bool WIDE = NT > 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto for_each = [&](auto&& f) {
    if constexpr (WIDE) {
      for (int n = wave; n < N; n += NT / 64) {
        const long base = ((long)n * C + c) * plane;
        for (int p = lane; p < plane; p += 64) f(base + p);
      }
    } else {
      for (long e = threadIdx.x; e < m; e += NT) {
        const int n = (int)(e / plane), p = (int)(e - (long)n * plane);
        f(((long)n * C + c) * plane + p);
      }
    }
  };
  // wide form, <= 64 images of <= 512 positions (every training-step launch): a thread's <= 32 elements stay in registers
  // across the three passes -- z is read once instead of three times, and the passes are no longer three dependent
  // round trips to memory
  constexpr int CI = 4, CP = 8;
  const bool cached = WIDE && N <= (NT / 64) * CI && plane <= 64 * CP;
  float zc[CI][CP];
  unsigned vm = 0;
  auto slot_idx = [&](int i, int jj) { return ((long)(wave + (NT / 64) * i) * C + c) * plane + lane + 64 * jj; };
  if (cached) {
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int jj = 0; jj < CP; ++jj) {
        const bool ok = wave + (NT / 64) * i < N && lane + 64 * jj < plane;
        zc[i][jj] = ok ? z[slot_idx(i, jj)] : 0.f;
        vm |= ok ? (1u << (i * CP + jj)) : 0u;
      }
  }
  auto for_cached = [&](auto&& f) {
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int jj = 0; jj < CP; ++jj)
        if ((vm >> (i * CP + jj)) & 1u) f(i, jj);
  };
  float s = 0.f;
  if (cached) for_cached([&](int i, int jj) { s += zc[i][jj]; });
  else for_each([&](long idx) { s += z[idx]; });
  const float mean = block_sum<NT>(s, sh) / (float)m;
  float q = 0.f;
  if (cached) for_cached([&](int i, int jj) { const float d = zc[i][jj] - mean; q += d * d; });
  else for_each([&](long idx) { const float d = z[idx] - mean; q += d * d; });
  const float var = block_sum<NT>(q, sh) / (float)m;
  const float istd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_o[c] = mean;
    istd_o[c] = istd;
    if (!(hold && *hold)) {  // (no running-average update from a pass the timeout flag has declared void)
      const float adjust = (float)((double)m / (m - 1 > 1 ? (double)(m - 1) : 1.0));
      avg_mean[c] = avg_mean[c] * decay + (1.f - decay) * mean;
      avg_var[c] = avg_var[c] * decay + ((1.f - decay) * adjust) * var;
    }
  }
  const float g = gamma[c], b = beta[c];
  if (cached) {
    for_cached([&](int i, int jj) {
      const float v = g * ((zc[i][jj] - mean) * istd) + b;
      y[slot_idx(i, jj)] = v >= 0.f ? v : slope * v;
    });
  } else {
    for_each([&](long idx) {
      const float v = g * ((z[idx] - mean) * istd) + b;
      y[idx] = v >= 0.f ? v : slope * v;
    });
  }
}


// Register-resident, BRANCH-FREE form of the two kernels for the launches of the training step (round 5): 1024 threads, one wavefront
// per image quadruple, CP slots of 64 positions per image (planes of <= 64 CP positions, <= 64 images, tensor < 2 GB).  The general
// kernels above guard every element of their cached path with `if (valid)`: hipcc then waits with vmcnt(0) inside every guarded block
// -- from the second block on for the PREVIOUS block's store -- and their fixed 8 slots per image are mostly empty on the 9 x 9 planes.
// Here every access is a raw buffer access (offset -1: reads zero / is dropped), the slot count is a template parameter (2: 9 x 9,
// 6: 18 x 18), the per-channel scalars are requested first, and nothing stands between the loads or between the stores.  Same
// element -> thread mapping and the same summation order as the general kernels (the results differ in the last bit where hipcc
// contracts a multiply-add in one form and not in the other).
template <int CP>
__global__ __launch_bounds__(1024) void bn_train_fwd_reg_kernel(const float* __restrict__ z, float* __restrict__ y, const float* gamma,
                                                                const float* beta, float* mean_o, float* istd_o, float* avg_mean,
                                                                float* avg_var, int N, int C, int plane, float eps, float decay, float slope,
                                                                const int* hold) {
  constexpr int NT = 1024, CI = 4;
  __shared__ float sh[NT / 64];
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbytes = (int)(4L * N * C * plane);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, nbytes, 0x00020000);
  const float g = gamma[c], b = beta[c];
  const bool upd = threadIdx.x == 0 && !(hold && *hold);  // (no running-average update from a pass the timeout flag has declared void)
  const float am0 = threadIdx.x == 0 ? avg_mean[c] : 0.f, av0 = threadIdx.x == 0 ? avg_var[c] : 0.f;
  auto zo = [&](int i, int jj) {   // byte offset of slot (i, jj), -1: no element
    const bool ok = wave + (NT / 64) * i < N && lane + 64 * jj < plane;
    return ok ? 4 * (((wave + (NT / 64) * i) * C + c) * plane + lane + 64 * jj) : -1;
  };
  float zc[CI][CP];
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) zc[i][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zo(i, jj), 0, 0));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) s += zc[i][jj];   // (absent slots hold zero)
  const float mean = block_sum<NT>(s, sh) / (float)m;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) {
      const float d = zc[i][jj] - mean;
      q += zo(i, jj) >= 0 ? d * d : 0.f;
    }
  const float var = block_sum<NT>(q, sh) / (float)m;
  const float istd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_o[c] = mean;
    istd_o[c] = istd;
    if (upd) {
      const float adjust = (float)((double)m / (m - 1 > 1 ? (double)(m - 1) : 1.0));
      avg_mean[c] = am0 * decay + (1.f - decay) * mean;
      avg_var[c] = av0 * decay + ((1.f - decay) * adjust) * var;
    }
  }
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) {
      const float v = g * ((zc[i][jj] - mean) * istd) + b;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v >= 0.f ? v : slope * v), ry, zo(i, jj), 0, 0);
    }
}

template <int CP>
__global__ __launch_bounds__(1024) void bn_train_bwd_reg_kernel(const float* __restrict__ z, const float* __restrict__ gh, const float* gamma,
                                                                const float* beta, const float* mean_i, const float* istd_i,
                                                                float* __restrict__ gz, float* ggamma, float* gbeta, int N, int C, int plane,
                                                                float slope) {
  constexpr int NT = 1024, CI = 4;
  __shared__ float sh[NT / 64];
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  const float mean = mean_i[c], istd = istd_i[c], g = gamma[c], b = beta[c];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbytes = (int)(4L * N * C * plane);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gh), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(gz, 0, nbytes, 0x00020000);
  auto zo = [&](int i, int jj) {
    const bool ok = wave + (NT / 64) * i < N && lane + 64 * jj < plane;
    return ok ? 4 * (((wave + (NT / 64) * i) * C + c) * plane + lane + 64 * jj) : -1;
  };
  float xc[CI][CP], gc[CI][CP];
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) {
      xc[i][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zo(i, jj), 0, 0));
      gc[i][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, zo(i, jj), 0, 0));
    }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) {
      const bool ok = zo(i, jj) >= 0;
      const float xh = ok ? (xc[i][jj] - mean) * istd : 0.f;
      const float yv = g * xh + b;
      const float gt = ok ? (yv >= 0.f ? gc[i][jj] : slope * gc[i][jj]) : 0.f;
      s1 += gt;
      s2 += gt * xh;
      xc[i][jj] = xh;
      gc[i][jj] = gt;
    }
  const float sg = block_sum<NT>(s1, sh);
  const float sgx = block_sum<NT>(s2, sh);
  if (threadIdx.x == 0) {
    atomicAdd(ggamma + c, sgx);  // atomics: the real- and fake-batch backward passes run concurrently
    atomicAdd(gbeta + c, sg);
  }
  const float k = g * istd, im = 1.f / (float)m;
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int jj = 0; jj < CP; ++jj)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, k * (gc[i][jj] - (sg + xc[i][jj] * sgx) * im)), ro, zo(i, jj), 0, 0);
}

// The same for the deep layers (planes of 4 x 4 .. 1 x 1, 256 - 512 channels): 256 threads, flat index e = thread + 256 k over (image,
// position), K elements per thread in registers (N * plane <= 256 K) instead of three dependent passes over memory.  Element -> thread
// mapping and summation order are those of the general 256-thread kernels.
template <int K>
__global__ __launch_bounds__(256) void bn_train_fwd_flat_kernel(const float* __restrict__ z, float* __restrict__ y, const float* gamma,
                                                                const float* beta, float* mean_o, float* istd_o, float* avg_mean,
                                                                float* avg_var, int N, int C, int plane, float eps, float decay, float slope,
                                                                const int* hold) {
  __shared__ float sh[4];
  const int c = blockIdx.x, m = N * plane;
  const int nbytes = (int)(4L * N * C * plane);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, nbytes, 0x00020000);
  const float g = gamma[c], b = beta[c];
  const bool upd = threadIdx.x == 0 && !(hold && *hold);
  const float am0 = threadIdx.x == 0 ? avg_mean[c] : 0.f, av0 = threadIdx.x == 0 ? avg_var[c] : 0.f;
  int zo[K];
  float zc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = threadIdx.x + 256 * k;
    const int n = e / plane, p = e - n * plane;
    zo[k] = e < m ? 4 * ((n * C + c) * plane + p) : -1;
    zc[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zo[k], 0, 0));
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += zc[k];
  const float mean = block_sum<256>(s, sh) / (float)m;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float d = zc[k] - mean;
    q += zo[k] >= 0 ? d * d : 0.f;
  }
  const float var = block_sum<256>(q, sh) / (float)m;
  const float istd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_o[c] = mean;
    istd_o[c] = istd;
    if (upd) {
      const float adjust = (float)((double)m / (m - 1 > 1 ? (double)(m - 1) : 1.0));
      avg_mean[c] = am0 * decay + (1.f - decay) * mean;
      avg_var[c] = av0 * decay + ((1.f - decay) * adjust) * var;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float v = g * ((zc[k] - mean) * istd) + b;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v >= 0.f ? v : slope * v), ry, zo[k], 0, 0);
  }
}

template <int K>
__global__ __launch_bounds__(256) void bn_train_bwd_flat_kernel(const float* __restrict__ z, const float* __restrict__ gh, const float* gamma,
                                                                const float* beta, const float* mean_i, const float* istd_i,
                                                                float* __restrict__ gz, float* ggamma, float* gbeta, int N, int C, int plane,
                                                                float slope) {
  __shared__ float sh[4];
  const int c = blockIdx.x, m = N * plane;
  const float mean = mean_i[c], istd = istd_i[c], g = gamma[c], b = beta[c];
  const int nbytes = (int)(4L * N * C * plane);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gh), 0, nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(gz, 0, nbytes, 0x00020000);
  int zo[K];
  float xc[K], gc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = threadIdx.x + 256 * k;
    const int n = e / plane, p = e - n * plane;
    zo[k] = e < m ? 4 * ((n * C + c) * plane + p) : -1;
    xc[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, zo[k], 0, 0));
    gc[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, zo[k], 0, 0));
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool ok = zo[k] >= 0;
    const float xh = ok ? (xc[k] - mean) * istd : 0.f;
    const float yv = g * xh + b;
    const float gt = ok ? (yv >= 0.f ? gc[k] : slope * gc[k]) : 0.f;
    s1 += gt;
    s2 += gt * xh;
    xc[k] = xh;
    gc[k] = gt;
  }
  const float sg = block_sum<256>(s1, sh);
  const float sgx = block_sum<256>(s2, sh);
  if (threadIdx.x == 0) {
    atomicAdd(ggamma + c, sgx);
    atomicAdd(gbeta + c, sg);
  }
  const float k2 = g * istd, im = 1.f / (float)m;
#pragma unroll
  for (int k = 0; k < K; ++k)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, k2 * (gc[k] - (sg + xc[k] * sgx) * im)), ro, zo[k], 0, 0);
}

// elements per thread of the flat register-resident form for (N, C, plane), 0: none
static int bn_flat_slots(int N, int C, int plane) {
  static const bool on = !(getenv("DBM_BN_REG") && atoi(getenv("DBM_BN_REG")) == 0);
  const long m = (long)N * plane;
  if (!on || (plane >= 64 && C <= 256) || 4L * m * C >= (1L << 31)) return 0;
  return m <= 256 ? 1 : m <= 1024 ? 4 : 0;
}

// which register-resident form serves (N, C, plane): its slots per image, 0: none (DBM_BN_REG=0: never -- A/B)
static int bn_reg_slots(int N, int C, int plane) {
  static const bool on = !(getenv("DBM_BN_REG") && atoi(getenv("DBM_BN_REG")) == 0);
  if (!on || N > 64 || plane < 64 || C > 256 || 4L * N * C * plane >= (1L << 31)) return 0;
  return plane <= 128 ? 2 : plane <= 384 ? 6 : 0;
}

void launch_bn_train_fwd(const float* z, float* y, const float* gamma, const float* beta, float* mean, float* inv_std,
                         float* avg_mean, float* avg_var, int N, int C, int plane, float eps, float decay, float slope,
                         hipStream_t s, const int* hold) {
  if (dbm_abl_skip() & 1) return;  // (libdbm_measure.so only)
  // (round 5 A/B: 256-thread workgroups everywhere -- easier to place beside other kernels -- cost 8.69-8.71 against 7.91 ms per step)
  const int cp = bn_reg_slots(N, C, plane);
  if (cp == 2)
    hipLaunchKernelGGL(bn_train_fwd_reg_kernel<2>, dim3(C), dim3(1024), 0, s, z, y, gamma, beta, mean, inv_std, avg_mean, avg_var, N, C,
                       plane, eps, decay, slope, hold);
  else if (cp == 6)
    hipLaunchKernelGGL(bn_train_fwd_reg_kernel<6>, dim3(C), dim3(1024), 0, s, z, y, gamma, beta, mean, inv_std, avg_mean, avg_var, N, C,
                       plane, eps, decay, slope, hold);
  else if (bn_flat_slots(N, C, plane) == 1)
    hipLaunchKernelGGL(bn_train_fwd_flat_kernel<1>, dim3(C), dim3(256), 0, s, z, y, gamma, beta, mean, inv_std, avg_mean, avg_var, N, C,
                       plane, eps, decay, slope, hold);
  else if (bn_flat_slots(N, C, plane) == 4)
    hipLaunchKernelGGL(bn_train_fwd_flat_kernel<4>, dim3(C), dim3(256), 0, s, z, y, gamma, beta, mean, inv_std, avg_mean, avg_var, N, C,
                       plane, eps, decay, slope, hold);
  else if (plane >= 64 && C <= 256)
    hipLaunchKernelGGL(bn_train_fwd_kernel<1024>, dim3(C), dim3(1024), 0, s, z, y, gamma, beta, mean, inv_std, avg_mean,
                       avg_var, N, C, plane, eps, decay, slope, hold);
  else
    hipLaunchKernelGGL(bn_train_fwd_kernel<256>, dim3(C), dim3(256), 0, s, z, y, gamma, beta, mean, inv_std, avg_mean,
                       avg_var, N, C, plane, eps, decay, slope, hold);
  DBM_HIP(hipGetLastError());
}

// Eval-mode BatchNorm (chainer.config.train = False, srgan_train.py:1228) as a per-channel affine map of the convolution's
// output: scale = gamma / sqrt(avg_var + eps), shift = beta - avg_mean * scale, for ALL layers of the discriminator in one launch;
// the convolutions then apply them in their epilogues (ConvDesc::ch_scale / bias) together with the LeakyReLU: an eval-mode
// pass has no BatchNorm launches of its own and never writes the pre-normalisation planes.
__global__ __launch_bounds__(256) void bn_eval_coeffs_kernel(BnEvalJobs jobs, float eps) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= jobs.total) return;
  int l = 0;
#pragma unroll
  for (int k = 1; k < BnEvalJobs::MAXL; ++k) l += (k < jobs.n && e >= jobs.start[k]) ? 1 : 0;
  const int c = e - jobs.start[l];
  const float sc = jobs.gamma[l][c] / sqrtf(jobs.avg_var[l][c] + eps);
  jobs.scale[e] = sc;
  jobs.shift[e] = jobs.beta[l][c] - jobs.avg_mean[l][c] * sc;
}
void launch_bn_eval_coeffs(const BnEvalJobs& jobs, float eps, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)((jobs.total + 255) / 256)), dim3(256), 0, s, jobs, eps);
  DBM_HIP(hipGetLastError());
}

template <int NT>
__global__ __launch_bounds__(NT) void bn_train_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gh,
                                                          const float* gamma, const float* beta, const float* mean_i,
                                                          const float* istd_i, float* __restrict__ gz, float* ggamma,
                                                          float* gbeta, int N, int C, int plane, float slope) {
  __shared__ float sh[NT / 64];
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  const float mean = mean_i[c], istd = istd_i[c], g = gamma[c], b = beta[c];
  constexpr bool WIDE = NT > 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto for_each = [&](auto&& f) {
    if constexpr (WIDE) {
      for (int n = wave; n < N; n += NT / 64) {
        const long base = ((long)n * C + c) * plane;
        for (int p = lane; p < plane; p += 64) f(base + p);
      }
    } else {
      for (long e = threadIdx.x; e < m; e += NT) {
        const int n = (int)(e / plane), p = (int)(e - (long)n * plane);
        f(((long)n * C + c) * plane + p);
      }
    }
  };
  // (wide form: x-hat and the masked gradient of a thread's <= 32 elements stay in registers between the two passes)
  constexpr int CI = 4, CP = 8;
  const bool cached = WIDE && N <= (NT / 64) * CI && plane <= 64 * CP;
  float xc[CI][CP], gc[CI][CP];
  unsigned vm = 0;
  auto slot_idx = [&](int i, int jj) { return ((long)(wave + (NT / 64) * i) * C + c) * plane + lane + 64 * jj; };
  float s1 = 0.f, s2 = 0.f;
  if (cached) {
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int jj = 0; jj < CP; ++jj) {
        const bool ok = wave + (NT / 64) * i < N && lane + 64 * jj < plane;
        float xh = 0.f, gt = 0.f;
        if (ok) {
          const long idx = slot_idx(i, jj);
          xh = (z[idx] - mean) * istd;
          const float yv = g * xh + b;
          gt = yv >= 0.f ? gh[idx] : slope * gh[idx];
          vm |= 1u << (i * CP + jj);
          s1 += gt;
          s2 += gt * xh;
        }
        xc[i][jj] = xh;
        gc[i][jj] = gt;
      }
  } else {
    for_each([&](long idx) {
      const float xh = (z[idx] - mean) * istd;
      const float yv = g * xh + b;
      const float gt = yv >= 0.f ? gh[idx] : slope * gh[idx];
      s1 += gt;
      s2 += gt * xh;
    });
  }
  const float sg = block_sum<NT>(s1, sh);
  const float sgx = block_sum<NT>(s2, sh);
  if (threadIdx.x == 0) {
    atomicAdd(ggamma + c, sgx);  // atomics: the real- and fake-batch backward passes run concurrently
    atomicAdd(gbeta + c, sg);
  }
  const float k = g * istd, im = 1.f / (float)m;
  if (cached) {
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int jj = 0; jj < CP; ++jj)
        if ((vm >> (i * CP + jj)) & 1u) gz[slot_idx(i, jj)] = k * (gc[i][jj] - (sg + xc[i][jj] * sgx) * im);
  } else {
    for_each([&](long idx) {
      const float xh = (z[idx] - mean) * istd;
      const float yv = g * xh + b;
      const float gt = yv >= 0.f ? gh[idx] : slope * gh[idx];
      gz[idx] = k * (gt - (sg + xh * sgx) * im);
    });
  }
}

void launch_bn_train_bwd(const float* z, const float* gh, const float* gamma, const float* beta, const float* mean,
                         const float* inv_std, float* gz, float* ggamma, float* gbeta, float* scratch, int N, int C,
                         int plane, float slope, hipStream_t s) {
  if (dbm_abl_skip() & 1) return;  // (libdbm_measure.so only)
  (void)scratch;
  const int cp = bn_reg_slots(N, C, plane);
  if (cp == 2)
    hipLaunchKernelGGL(bn_train_bwd_reg_kernel<2>, dim3(C), dim3(1024), 0, s, z, gh, gamma, beta, mean, inv_std, gz, ggamma, gbeta, N, C,
                       plane, slope);
  else if (cp == 6)
    hipLaunchKernelGGL(bn_train_bwd_reg_kernel<6>, dim3(C), dim3(1024), 0, s, z, gh, gamma, beta, mean, inv_std, gz, ggamma, gbeta, N, C,
                       plane, slope);
  else if (bn_flat_slots(N, C, plane) == 1)
    hipLaunchKernelGGL(bn_train_bwd_flat_kernel<1>, dim3(C), dim3(256), 0, s, z, gh, gamma, beta, mean, inv_std, gz, ggamma, gbeta, N, C,
                       plane, slope);
  else if (bn_flat_slots(N, C, plane) == 4)
    hipLaunchKernelGGL(bn_train_bwd_flat_kernel<4>, dim3(C), dim3(256), 0, s, z, gh, gamma, beta, mean, inv_std, gz, ggamma, gbeta, N, C,
                       plane, slope);
  else if (plane >= 64 && C <= 256)
    hipLaunchKernelGGL(bn_train_bwd_kernel<1024>, dim3(C), dim3(1024), 0, s, z, gh, gamma, beta, mean, inv_std, gz, ggamma,
                       gbeta, N, C, plane, slope);
  else
    hipLaunchKernelGGL(bn_train_bwd_kernel<256>, dim3(C), dim3(256), 0, s, z, gh, gamma, beta, mean, inv_std, gz, ggamma,
                       gbeta, N, C, plane, slope);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// sync_batch_stats: BatchNorm over the GLOBAL batch of a data-parallel run.  Each pass is cut where the
// per-channel sums cross ranks: local statistics -> all-reduce (the caller's hook) -> apply.  Local means / M2 are
// combined Chan-style (sum of means, of M2 and of squared means: equal counts per rank), so the variance keeps the
// two-pass accuracy of the single-process kernel.
// ----------------------------------------------------------------------------------------------
#define BN_SYNC_FOR_EACH(body)                                              \
  for (long e = threadIdx.x; e < m; e += 256) {                             \
    const int n = (int)(e / plane), p = (int)(e - (long)n * plane);         \
    const long idx = ((long)n * C + c) * plane + p;                         \
    body                                                                    \
  }

__global__ __launch_bounds__(256) void bn_sync_stats_kernel(const float* __restrict__ z, float* buf, int N, int C, int plane) {
  __shared__ float sh[4];
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  float s = 0.f;
  BN_SYNC_FOR_EACH(s += z[idx];)
  const float mean = block_sum<256>(s, sh) / (float)m;
  float q = 0.f;
  BN_SYNC_FOR_EACH(const float d = z[idx] - mean; q += d * d;)
  const float M2 = block_sum<256>(q, sh);
  if (threadIdx.x == 0) { buf[c] = mean; buf[C + c] = M2; buf[2 * C + c] = mean * mean; }
}

__global__ __launch_bounds__(256) void bn_sync_fwd_apply_kernel(const float* __restrict__ z, float* __restrict__ y,
                                                                const float* gamma, const float* beta, const float* buf,
                                                                float* mean_o, float* istd_o, float* avg_mean, float* avg_var,
                                                                int N, int C, int plane, int world, float eps, float decay,
                                                                float slope, const int* hold) {
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  const double mg = (double)m * world;
  const float mean = buf[c] / (float)world;
  const float M2 = buf[C + c] + (float)m * (buf[2 * C + c] - (float)world * mean * mean);
  const float var = fmaxf(M2, 0.f) / (float)mg;
  const float istd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_o[c] = mean;
    istd_o[c] = istd;
    if (!(hold && *hold)) {
      const float adjust = (float)(mg / (mg - 1 > 1 ? mg - 1 : 1.0));
      avg_mean[c] = avg_mean[c] * decay + (1.f - decay) * mean;
      avg_var[c] = avg_var[c] * decay + ((1.f - decay) * adjust) * var;
    }
  }
  const float g = gamma[c], b = beta[c];
  BN_SYNC_FOR_EACH(const float v = g * ((z[idx] - mean) * istd) + b; y[idx] = v >= 0.f ? v : slope * v;)
}

__global__ __launch_bounds__(256) void bn_sync_bwd_sums_kernel(const float* __restrict__ z, const float* __restrict__ gh,
                                                               const float* gamma, const float* beta, const float* mean_i,
                                                               const float* istd_i, float* buf, float* ggamma, float* gbeta,
                                                               int N, int C, int plane, float slope) {
  __shared__ float sh[4];
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  const float mean = mean_i[c], istd = istd_i[c], g = gamma[c], b = beta[c];
  float s1 = 0.f, s2 = 0.f;
  BN_SYNC_FOR_EACH(const float xh = (z[idx] - mean) * istd; const float yv = g * xh + b;
                   const float gt = yv >= 0.f ? gh[idx] : slope * gh[idx]; s1 += gt; s2 += gt * xh;)
  const float sg = block_sum<256>(s1, sh);
  const float sgx = block_sum<256>(s2, sh);
  if (threadIdx.x == 0) {
    buf[c] = sg; buf[C + c] = sgx;
    atomicAdd(ggamma + c, sgx);  // this rank's share: the gradient all-reduce sums the ranks
    atomicAdd(gbeta + c, sg);
  }
}

__global__ __launch_bounds__(256) void bn_sync_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ gh,
                                                                const float* gamma, const float* beta, const float* mean_i,
                                                                const float* istd_i, const float* buf, float* __restrict__ gz,
                                                                int N, int C, int plane, int world, float slope) {
  const int c = blockIdx.x;
  const long m = (long)N * plane;
  const float mean = mean_i[c], istd = istd_i[c], g = gamma[c], b = beta[c];
  const float sg = buf[c], sgx = buf[C + c];
  const float k = g * istd, im = 1.f / ((float)m * (float)world);
  BN_SYNC_FOR_EACH(const float xh = (z[idx] - mean) * istd; const float yv = g * xh + b;
                   const float gt = yv >= 0.f ? gh[idx] : slope * gh[idx]; gz[idx] = k * (gt - (sg + xh * sgx) * im);)
}

void launch_bn_sync_stats(const float* z, float* buf, int N, int C, int plane, hipStream_t s) {
  hipLaunchKernelGGL(bn_sync_stats_kernel, dim3(C), dim3(256), 0, s, z, buf, N, C, plane);
  DBM_HIP(hipGetLastError());
}
void launch_bn_sync_fwd_apply(const float* z, float* y, const float* gamma, const float* beta, const float* buf, float* mean,
                              float* inv_std, float* avg_mean, float* avg_var, int N, int C, int plane, int world, float eps,
                              float decay, float slope, hipStream_t s, const int* hold) {
  hipLaunchKernelGGL(bn_sync_fwd_apply_kernel, dim3(C), dim3(256), 0, s, z, y, gamma, beta, buf, mean, inv_std, avg_mean, avg_var,
                     N, C, plane, world, eps, decay, slope, hold);
  DBM_HIP(hipGetLastError());
}
void launch_bn_sync_bwd_sums(const float* z, const float* gh, const float* gamma, const float* beta, const float* mean,
                             const float* inv_std, float* buf, float* ggamma, float* gbeta, int N, int C, int plane, float slope,
                             hipStream_t s) {
  hipLaunchKernelGGL(bn_sync_bwd_sums_kernel, dim3(C), dim3(256), 0, s, z, gh, gamma, beta, mean, inv_std, buf, ggamma, gbeta, N,
                     C, plane, slope);
  DBM_HIP(hipGetLastError());
}
void launch_bn_sync_bwd_apply(const float* z, const float* gh, const float* gamma, const float* beta, const float* mean,
                              const float* inv_std, const float* buf, float* gz, int N, int C, int plane, int world, float slope,
                              hipStream_t s) {
  hipLaunchKernelGGL(bn_sync_bwd_apply_kernel, dim3(C), dim3(256), 0, s, z, gh, gamma, beta, mean, inv_std, buf, gz, N, C, plane,
                     world, slope);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// L.Linear head of the discriminator (srgan_train.py:646-647, 693-696).  51 300 + 101 parameters.
// ----------------------------------------------------------------------------------------------
// one wavefront per output element: both rows are read coalesced, then a shuffle reduction
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ y, int N,
                                                         int K, int O, int act, float slope) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= N * O) return;
  const int lane = threadIdx.x & 63;
  const int n = e / O, o = e - n * O;
  const float* xr = x + (long)n * K;
  const float* wr = W + (long)o * K;
  float a = 0.f;
  for (int k = lane; k < K; k += 64) a = fmaf(xr[k], wr[k], a);
  for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s, 64);
  if (lane == 0) {
    float v = a + b[o];
    if (act) v = v >= 0.f ? v : slope * v;
    y[e] = v;
  }
}

void launch_linear_fwd(const float* x, const float* W, const float* b, float* y, int N, int K, int O, int act,
                       float slope, hipStream_t s) {
  if (dbm_abl_skip() & 32) return;  // (libdbm_measure.so only)
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((N * O + 3) / 4), dim3(256), 0, s, x, W, b, y, N, K, O, act, slope);
  DBM_HIP(hipGetLastError());
}

__device__ __forceinline__ float gyz_of(const float* gy, const float* y_act, int idx, float slope) {
  const float g = gy[idx];
  return (y_act == nullptr || y_act[idx] >= 0.f) ? g : slope * g;
}

// STAGED: the masked output gradient gz = lrelu'(y) * gy (N x O floats) is read once per workgroup into LDS (coalesced, every request in
// flight at once) instead of two guarded loads per multiply-add -- those were O (or N) dependent round trips per thread: 20 us for the
// 64 x 512 x 100 layer at the head of both discriminator backward passes.
template <bool STAGED>
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ gy, const float* __restrict__ y_act,
                                                         float* gx, float* gW, float* gb, int N, int K, int O,
                                                         float slope) {
  extern __shared__ float gzs[];
  if (STAGED) {
    for (int i = threadIdx.x; i < N * O; i += 256) gzs[i] = gyz_of(gy, y_act, i, slope);
    __syncthreads();
  }
  auto gz = [&](int idx) { return STAGED ? gzs[idx] : gyz_of(gy, y_act, idx, slope); };
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int nx = N * K, nw = O * K;
  if (e < nx) {  // gx[n][k]
    const int n = e / K, k = e - n * K;
    float a = 0.f;
#pragma unroll 10
    for (int o = 0; o < O; ++o) a = fmaf(gz(n * O + o), W[(long)o * K + k], a);
    gx[e] = a;
  } else if (e < nx + nw) {  // gW[o][k]
    const int q = e - nx;
    const int o = q / K, k = q - o * K;
    float a = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) a = fmaf(gz(n * O + o), x[(long)n * K + k], a);
    atomicAdd(gW + q, a);
  } else if (e < nx + nw + O) {
    const int o = e - nx - nw;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += gz(n * O + o);
    atomicAdd(gb + o, a);
  }
}

void launch_linear_bwd(const float* x, const float* W, const float* gy, const float* y_act, float* gx, float* gW,
                       float* gb, int N, int K, int O, float slope, hipStream_t s) {
  if (dbm_abl_skip() & 32) return;  // (libdbm_measure.so only)
  const int total = N * K + O * K + O;
  const size_t lds = sizeof(float) * (size_t)N * O;
  if (lds <= 48 * 1024)
    hipLaunchKernelGGL(linear_bwd_kernel<true>, dim3((total + 255) / 256), dim3(256), lds, s, x, W, gy, y_act, gx, gW, gb, N, K, O, slope);
  else
    hipLaunchKernelGGL(linear_bwd_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, s, x, W, gy, y_act, gx, gW, gb, N, K, O, slope);
  DBM_HIP(hipGetLastError());
}

// ---- the discriminator's head as ONE launch per pass (round 6) ----
// linear_1 -> LeakyReLU -> linear_2 (:693-696) used to be two launches of linear_fwd_kernel at the end of every discriminator forward
// (three per iteration) and two of linear_bwd_kernel at the head of every backward pass, i.e. of both serial chains the D-step waits for;
// inside the iteration a launch of that size costs 20-40 us of queueing whatever it computes.  Same arithmetic in the same order as the
// two-launch form (every sum below is linear_fwd_kernel's / linear_bwd_kernel's own): results are bitwise the same.
// Forward: one workgroup per image; wavefront w computes outputs w, w + 4, ... of linear_1 (lanes stride the 512 inputs, shuffle tree),
// the row goes to memory (the backward pass reads it) and to LDS, wavefront 0 then forms the logit.
template <int K, int NW>
__global__ __launch_bounds__(64 * NW) void disc_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W1,
                                                                const float* __restrict__ b1, const float* __restrict__ W2,
                                                                const float* __restrict__ b2, float* __restrict__ l1,
                                                                float* __restrict__ logits, int O, float slope) {
  // (first version: four wavefronts, 25 outputs each, one after the other -- 25 dependent L2 round trips per wavefront: 7.75-7.78
  //  against 7.65-7.67 ms per step for the two launches it replaced, profiles/r6/ab_disc_head.txt.  Now sixteen wavefronts per image,
  //  the image's row in registers, and the weight rows of a wavefront's outputs requested four at a time.)
  __shared__ float row[256];
  constexpr int KP = K / 64;
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xr = x + (long)n * K;
  float xv[KP];
#pragma unroll
  for (int i = 0; i < KP; ++i) xv[i] = xr[lane + 64 * i];
  for (int o0 = wave; o0 < O; o0 += 4 * NW) {
    float wv[4][KP];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * NW < O ? o0 + u * NW : o0;
#pragma unroll
      for (int i = 0; i < KP; ++i) wv[u][i] = W1[(long)o * K + lane + 64 * i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * NW;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < KP; ++i) a = fmaf(xv[i], wv[u][i], a);   // (k = lane, lane + 64, ...: linear_fwd_kernel's order)
      for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s, 64);
      if (lane == 0 && o < O) {
        float v = a + b1[o];
        v = v >= 0.f ? v : slope * v;
        l1[(long)n * O + o] = v;
        row[o] = v;
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    float a = 0.f;
    for (int k = lane; k < O; k += 64) a = fmaf(row[k], W2[k], a);
    for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s, 64);
    if (lane == 0) logits[n] = a + b2[0];
  }
}

void launch_disc_head_fwd(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, float* l1, float* logits,
                          int N, int K, int O, float slope, hipStream_t s) {
  if (dbm_abl_skip() & 32) return;  // (libdbm_measure.so only)
  DBM_CHECK(O <= 256 && K == 512, "discriminator head: linear_1 is 512 -> at most 256");
  hipLaunchKernelGGL((disc_head_fwd_kernel<512, 16>), dim3(N), dim3(1024), 0, s, x, W1, b1, W2, b2, l1, logits, O, slope);
  DBM_HIP(hipGetLastError());
}

// Backward: linear_2's input gradient g_l1[n][o] = glogit[n] * W2[o] needs no reduction, so every workgroup forms the masked gradient
// gz[n][o] = lrelu'(l1[n][o]) * g_l1[n][o] in LDS itself (N x O floats) and then runs linear_bwd_kernel's element mapping for linear_1
// (gx, gW1, gb1) plus O + 1 more elements for linear_2's own gW2 / gb2.  g_l1 is not written: nothing else reads it.
__global__ __launch_bounds__(256) void disc_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W1,
                                                            const float* __restrict__ W2, const float* __restrict__ glogits,
                                                            const float* __restrict__ l1, float* gx, float* gW1, float* gb1,
                                                            float* gW2, float* gb2, int N, int K, int O, float slope) {
  extern __shared__ float gzs[];
  for (int i = threadIdx.x; i < N * O; i += 256) {
    const int n = i / O, o = i - n * O;
    const float g = fmaf(glogits[n], W2[o], 0.f);   // (linear_bwd_kernel's gx of linear_2: one multiply-add onto zero)
    gzs[i] = l1[i] >= 0.f ? g : slope * g;
  }
  __syncthreads();
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int nx = N * K, nw = O * K;
  if (e < nx) {  // gx[n][k]
    const int n = e / K, k = e - n * K;
    float a = 0.f;
#pragma unroll 10
    for (int o = 0; o < O; ++o) a = fmaf(gzs[n * O + o], W1[(long)o * K + k], a);
    gx[e] = a;
  } else if (e < nx + nw) {  // gW1[o][k]
    const int q = e - nx;
    const int o = q / K, k = q - o * K;
    float a = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) a = fmaf(gzs[n * O + o], x[(long)n * K + k], a);
    atomicAdd(gW1 + q, a);
  } else if (e < nx + nw + O) {  // gb1[o]
    const int o = e - nx - nw;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += gzs[n * O + o];
    atomicAdd(gb1 + o, a);
  } else if (e < nx + nw + 2 * O) {  // gW2[o] = sum_n glogit[n] * l1[n][o]
    const int o = e - nx - nw - O;
    float a = 0.f;
#pragma unroll 8
    for (int n = 0; n < N; ++n) a = fmaf(glogits[n], l1[(long)n * O + o], a);
    atomicAdd(gW2 + o, a);
  } else if (e == nx + nw + 2 * O) {  // gb2
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += glogits[n];
    atomicAdd(gb2, a);
  }
}

bool disc_head_bwd_fused_ok(int N, int O) { return sizeof(float) * (size_t)N * O <= 48 * 1024; }
void launch_disc_head_bwd(const float* x, const float* W1, const float* W2, const float* glogits, const float* l1, float* gx, float* gW1,
                          float* gb1, float* gW2, float* gb2, int N, int K, int O, float slope, hipStream_t s) {
  if (dbm_abl_skip() & 32) return;  // (libdbm_measure.so only)
  const int total = N * K + O * K + 2 * O + 1;
  hipLaunchKernelGGL(disc_head_bwd_kernel, dim3((total + 255) / 256), dim3(256), sizeof(float) * (size_t)N * O, s, x, W1, W2, glogits, l1, gx,
                     gW1, gb1, gW2, gb2, N, K, O, slope);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// calculate_discriminator_loss (srgan_train.py:960-1009) + F.binary_accuracy (:1156-1158)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float sce_elem(float x, float t) {
  return -(x * (t - (x >= 0.f ? 1.f : 0.f)) - log1pf(expf(-fabsf(x))));
}

// tr_arr / tf_arr (may be null): per-sample int32 targets of the two F.sigmoid_cross_entropy calls (-1 = ignored; the
// reference's signature accepts any int array, srgan_train.py:995-1004); null = the constant tr / tf for every sample.
// Each call normalises by max(count(t != -1), 1) (normalize=True).
__global__ __launch_bounds__(256) void ragan_loss_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                         int N, float tr, float tf, const int* __restrict__ tr_arr,
                                                         const int* __restrict__ tf_arr, float* out, float* g_real,
                                                         float* g_fake) {
  __shared__ float sh[4];
  float sr = 0.f, sf = 0.f, c1 = 0.f, c2 = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    sr += real[i];
    sf += fake[i];
    c1 += (tr_arr && tr_arr[i] == -1) ? 0.f : 1.f;
    c2 += (tf_arr && tf_arr[i] == -1) ? 0.f : 1.f;
  }
  const float mr = block_sum_256(sr, sh) / (float)N;
  const float mf = block_sum_256(sf, sh) / (float)N;
  const float n1 = fmaxf(block_sum_256(c1, sh), 1.f), n2 = fmaxf(block_sum_256(c2, sh), 1.f);
  float l1 = 0.f, l2 = 0.f, acc = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float xr = real[i] - mf, xf = fake[i] - mr;
    const float t1 = tr_arr ? (float)tr_arr[i] : tr, t2 = tf_arr ? (float)tf_arr[i] : tf;
    if (t1 != -1.f) { l1 += sce_elem(xr, t1); s1 += (1.f / (1.f + expf(-xr)) - t1) / n1; }
    if (t2 != -1.f) { l2 += sce_elem(xf, t2); s2 += (1.f / (1.f + expf(-xf)) - t2) / n2; }
    acc += (real[i] >= 0.f ? 1.f : 0.f) + (fake[i] >= 0.f ? 0.f : 1.f);
  }
  const float L = block_sum_256(l1, sh) / n1 + block_sum_256(l2, sh) / n2;
  const float A = block_sum_256(acc, sh) / (float)(2 * N);
  const float S1 = block_sum_256(s1, sh);
  const float S2 = block_sum_256(s2, sh);
  if (threadIdx.x == 0) {
    out[0] = L;
    out[1] = A;
  }
  if (g_real && g_fake) {
    for (int i = threadIdx.x; i < N; i += 256) {
      const float xr = real[i] - mf, xf = fake[i] - mr;
      const float t1 = tr_arr ? (float)tr_arr[i] : tr, t2 = tf_arr ? (float)tf_arr[i] : tf;
      g_real[i] = (t1 != -1.f ? (1.f / (1.f + expf(-xr)) - t1) / n1 : 0.f) - S2 / (float)N;
      g_fake[i] = (t2 != -1.f ? (1.f / (1.f + expf(-xf)) - t2) / n2 : 0.f) - S1 / (float)N;
    }
  }
}

// sync_batch_stats form of the relativistic-average loss: the means of the other class's logits are GLOBAL-batch means
// (srgan_train.py:995-1004 at the global batch).  buf[0..1] = sums of the logits (all-reduced by the caller), then
// buf[2..3] = this rank's sums of the sigmoid residuals (all-reduced), then the gradients.
__global__ __launch_bounds__(256) void ragan_sync_sums_kernel(const float* real, const float* fake, int N, float* buf) {
  __shared__ float sh[4];
  float sr = 0.f, sf = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) { sr += real[i]; sf += fake[i]; }
  const float a = block_sum_256(sr, sh), b = block_sum_256(sf, sh);
  if (threadIdx.x == 0) { buf[0] = a; buf[1] = b; }
}
__global__ __launch_bounds__(256) void ragan_sync_loss_kernel(const float* real, const float* fake, int N, int world, float tr,
                                                              float tf, float* buf, float* out) {
  __shared__ float sh[4];
  const float ng = (float)N * (float)world;
  const float mr = buf[0] / ng, mf = buf[1] / ng;
  float l = 0.f, acc = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float xr = real[i] - mf, xf = fake[i] - mr;
    l += sce_elem(xr, tr) + sce_elem(xf, tf);
    acc += (real[i] >= 0.f ? 1.f : 0.f) + (fake[i] >= 0.f ? 0.f : 1.f);
    s1 += (1.f / (1.f + expf(-xr)) - tr) / (float)N;
    s2 += (1.f / (1.f + expf(-xf)) - tf) / (float)N;
  }
  const float L = block_sum_256(l, sh) / (float)N;
  const float A = block_sum_256(acc, sh) / (float)(2 * N);
  const float S1 = block_sum_256(s1, sh), S2 = block_sum_256(s2, sh);
  if (threadIdx.x == 0) { out[0] = L; out[1] = A; buf[2] = S1; buf[3] = S2; }
}
__global__ __launch_bounds__(256) void ragan_sync_grad_kernel(const float* real, const float* fake, int N, int world, float tr,
                                                              float tf, const float* buf, float* g_real, float* g_fake) {
  const float ng = (float)N * (float)world;
  const float mr = buf[0] / ng, mf = buf[1] / ng;
  const float S1 = buf[2] / (float)world, S2 = buf[3] / (float)world;  // mean over ranks of the per-rank sums
  for (int i = threadIdx.x; i < N; i += 256) {
    const float xr = real[i] - mf, xf = fake[i] - mr;
    g_real[i] = (1.f / (1.f + expf(-xr)) - tr) / (float)N - S2 / (float)N;
    g_fake[i] = (1.f / (1.f + expf(-xf)) - tf) / (float)N - S1 / (float)N;
  }
}
void launch_ragan_sync_sums(const float* real, const float* fake, int N, float* buf, hipStream_t s) {
  hipLaunchKernelGGL(ragan_sync_sums_kernel, dim3(1), dim3(256), 0, s, real, fake, N, buf);
  DBM_HIP(hipGetLastError());
}
void launch_ragan_sync_loss(const float* real, const float* fake, int N, int world, int real_target, int fake_target, float* buf,
                            float* out, hipStream_t s) {
  hipLaunchKernelGGL(ragan_sync_loss_kernel, dim3(1), dim3(256), 0, s, real, fake, N, world, (float)real_target,
                     (float)fake_target, buf, out);
  DBM_HIP(hipGetLastError());
}
void launch_ragan_sync_grad(const float* real, const float* fake, int N, int world, int real_target, int fake_target,
                            const float* buf, float* g_real, float* g_fake, hipStream_t s) {
  hipLaunchKernelGGL(ragan_sync_grad_kernel, dim3(1), dim3(256), 0, s, real, fake, N, world, (float)real_target,
                     (float)fake_target, buf, g_real, g_fake);
  DBM_HIP(hipGetLastError());
}

void launch_ragan_loss(const float* real, const float* fake, int N, int real_target, int fake_target, float* out,
                       float* g_real, float* g_fake, hipStream_t s, const int* real_targets, const int* fake_targets) {
  hipLaunchKernelGGL(ragan_loss_kernel, dim3(1), dim3(256), 0, s, real, fake, N, (float)real_target, (float)fake_target,
                     real_targets, fake_targets, out, g_real, g_fake);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// Generator loss terms (srgan_train.py:841-902): L1, topographic (4x4 mean vs BEDMAP2 interior), SSIM
// (9x9 separable window, valid positions) and their gradient w.r.t. y_pred.  One workgroup per tile;
// everything for a 36x36 tile lives in LDS.
// ----------------------------------------------------------------------------------------------
#define SSIM_K 9
__global__ __launch_bounds__(256) void gen_loss_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                       const float* __restrict__ X, int N, int H, int W, float cw,
                                                       float tw, float sw, const float* __restrict__ win, float* sums,
                                                       float* __restrict__ gy) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float sh[4];
  __shared__ float g1[SSIM_K];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int OH = H - (SSIM_K - 1), OW = W - (SSIM_K - 1);
  const int HW = H * W, HO = H * OW, OO = OH * OW;
  float* sY = sm;
  float* sT = sY + HW;
  float* hb = sT + HW;   // 5 x [H][OW] horizontal passes, later 3 x [H][OW] transposed-vertical passes
  float* cb = hb + 5 * HO;  // 3 x [OH][OW] coefficient maps
  if (tid < SSIM_K) g1[tid] = win[tid];
  const float* yn = y + (long)n * HW;
  const float* tn = t + (long)n * HW;
  float l1 = 0.f, sq = 0.f, tsum = 0.f, ysum = 0.f;
  for (int e = tid; e < HW; e += 256) {
    const float a = yn[e], b = tn[e];
    sY[e] = a;
    sT[e] = b;
    l1 += fabsf(a - b);
    sq += (a - b) * (a - b);
    tsum += b;
    ysum += a;
  }
  // Variances / covariances are shift invariant: work on (y - cy, t - ct) with the tile means so that
  // E[x^2] - mu^2 does not cancel catastrophically in fp32 on un-normalised elevations (metres).
  const float shift_t = block_sum_256(tsum, sh) / (float)HW;
  const float shift = block_sum_256(ysum, sh) / (float)HW;  // shift of y (also used by the pooled means below)
  for (int e = tid; e < HW; e += 256) {
    sY[e] -= shift;
    sT[e] -= shift_t;
  }
  __syncthreads();
  for (int e = tid; e < HO; e += 256) {
    const int i = e / OW, j = e - i * OW;
    float h1 = 0.f, h2 = 0.f, h11 = 0.f, h22 = 0.f, h12 = 0.f;
#pragma unroll
    for (int b = 0; b < SSIM_K; ++b) {
      const float w = g1[b], a = sY[i * W + j + b], c = sT[i * W + j + b];
      h1 += w * a; h2 += w * c; h11 += w * (a * a); h22 += w * (c * c); h12 += w * (a * c);
    }
    hb[e] = h1; hb[HO + e] = h2; hb[2 * HO + e] = h11; hb[3 * HO + e] = h22; hb[4 * HO + e] = h12;
  }
  __syncthreads();
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float ss = 0.f;
  for (int e = tid; e < OO; e += 256) {
    const int i = e / OW, j = e - i * OW;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int a = 0; a < SSIM_K; ++a) {
      const float w = g1[a];
      const int q = (i + a) * OW + j;
      m1 += w * hb[q]; m2 += w * hb[HO + q]; e11 += w * hb[2 * HO + q]; e22 += w * hb[3 * HO + q]; e12 += w * hb[4 * HO + q];
    }
    const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;  // m1, m2: means of the shifted data
    const float u1 = m1 + shift, u2 = m2 + shift_t;                              // true window means
    const float A1 = 2.f * u1 * u2 + C1, A2 = 2.f * s12 + C2, B1 = u1 * u1 + u2 * u2 + C1, B2 = s11 + s22 + C2;
    const float sv = (A1 * A2) / (B1 * B2);
    ss += sv;
    cb[e] = sv * (2.f * u2 / A1 - 2.f * m2 / A2 - 2.f * u1 / B1 + 2.f * m1 / B2);
    cb[OO + e] = -sv / B2;
    cb[2 * OO + e] = 2.f * sv / A2;
  }
  // topographic term: 4x4 block means vs X[:, :, 1:-1, 1:-1]
  const int PH = X ? H / 4 : 0, PW = X ? W / 4 : 0, XW = PW + 2;
  float tp = 0.f;
  for (int e = tid; e < PH * PW; e += 256) {
    const int pi = e / PW, pj = e - pi * PW;
    float a = 0.f;
    for (int u = 0; u < 4; ++u)
      for (int v = 0; v < 4; ++v) a += sY[(4 * pi + u) * W + 4 * pj + v];
    const float dlt = (a * (1.f / 16.f) + shift) - X[((long)n * (PH + 2) + pi + 1) * XW + pj + 1];
    tp += fabsf(dlt);
  }
  const float L1 = block_sum_256(l1, sh);
  const float SQ = block_sum_256(sq, sh);
  const float SS = block_sum_256(ss, sh);
  const float TP = block_sum_256(tp, sh);
  if (tid == 0) {  // per-tile partial sums: the finishing kernels add them in tile order (no atomics: reproducible)
    sums[4 * n + 0] = L1;
    sums[4 * n + 1] = TP;
    sums[4 * n + 2] = SS;
    sums[4 * n + 3] = SQ;
  }
  if (gy == nullptr) return;
  __syncthreads();
  // transposed vertical pass: vb_k[i][j] = sum_a g[a] * c_k[i-a][j]
  for (int e = tid; e < HO; e += 256) {
    const int i = e / OW, j = e - i * OW;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int a = 0; a < SSIM_K; ++a) {
      const int r = i - a;
      if (r >= 0 && r < OH) {
        const float w = g1[a];
        v0 += w * cb[r * OW + j]; v1 += w * cb[OO + r * OW + j]; v2 += w * cb[2 * OO + r * OW + j];
      }
    }
    hb[e] = v0; hb[HO + e] = v1; hb[2 * HO + e] = v2;
  }
  __syncthreads();
  const float k_l1 = cw / ((float)N * HW);
  const float k_tp = X ? tw / (16.f * (float)N * PH * PW) : 0.f;
  const float k_ss = sw / ((float)N * OO);
  for (int e = tid; e < HW; e += 256) {
    const int i = e / W, j = e - i * W;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int b = 0; b < SSIM_K; ++b) {
      const int c = j - b;
      if (c >= 0 && c < OW) {
        const float w = g1[b];
        s0 += w * hb[i * OW + c]; s1 += w * hb[HO + i * OW + c]; s2 += w * hb[2 * HO + i * OW + c];
      }
    }
    const float a = sY[e], b = sT[e];
    const float dss = s0 + 2.f * a * s1 + b * s2;
    const float d = yn[e] - tn[e];  // the un-shifted difference decides the sign of the L1 gradient
    float g = k_l1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) - k_ss * dss;
    const int pi = i >> 2, pj = j >> 2;
    if (pi < PH && pj < PW) {
      float pm = 0.f;
      for (int u = 0; u < 4; ++u)
        for (int v = 0; v < 4; ++v) pm += sY[(4 * pi + u) * W + 4 * pj + v];
      const float dlt = (pm * (1.f / 16.f) + shift) - X[((long)n * (PH + 2) + pi + 1) * XW + pj + 1];
      g += k_tp * (dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f));
    }
    gy[(long)n * HW + e] = g;
  }
}

void launch_gen_loss(const float* y, const float* t, const float* X, int N, int H, int W, float cw, float tw, float sw,
                     const float* win1d, float* sums, float* gy, hipStream_t s) {
  DBM_CHECK(H >= SSIM_K && W >= SSIM_K, "gen_loss: tile must be at least 9x9");
  DBM_CHECK(X == nullptr || (H % 4 == 0 && W % 4 == 0), "gen_loss: topographic term needs sides that are multiples of 4");
  const int OH = H - 8, OW = W - 8;
  const size_t lds = sizeof(float) * ((size_t)2 * H * W + 5 * (size_t)H * OW + 3 * (size_t)OH * OW);
  DBM_CHECK(lds <= 150 * 1024, "gen_loss: tile too large for the LDS-resident loss kernel");
  static bool attr_set = false;
  if (!attr_set) {
    DBM_HIP(hipFuncSetAttribute((const void*)gen_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(gen_loss_kernel, dim3(N), dim3(256), lds, s, y, t, X, N, H, W, cw, tw, sw, win1d, sums, gy);
  DBM_HIP(hipGetLastError());
}

// ssim_loss_func(y_pred, y_true, window_size, stride) for OTHER windows than the loss's 9 / 1 (srgan_train.py:932-956 passes
// both through to ssim.functions.ssim_loss): metric only, one workgroup per image, the two mean-shifted images in LDS when
// they fit (else read from memory), every output window by one thread.  Window = normalised gaussian(sigma 1.5) of
// `ws` taps centred at ws / 2 (pytorch-ssim lineage) or uniform; valid windows only; sums[4 n + 2] = sum of the SSIM map.
__global__ __launch_bounds__(256) void ssim_general_kernel(const float* __restrict__ y, const float* __restrict__ t, int H, int W,
                                                           int ws, int stride, int uniform, int in_lds, float* sums) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float sh[4];
  __shared__ float g1[64];
  const int n = blockIdx.x, tid = threadIdx.x, HW = H * W;
  const float* yn = y + (long)n * HW;
  const float* tn = t + (long)n * HW;
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < ws; ++i) {
      const float d = (float)(i - ws / 2);
      g1[i] = uniform ? 1.f : expf(-(d * d) / (2.f * 1.5f * 1.5f));
      s += g1[i];
    }
    for (int i = 0; i < ws; ++i) g1[i] /= s;
  }
  float ys = 0.f, ts = 0.f;
  for (int e = tid; e < HW; e += 256) { ys += yn[e]; ts += tn[e]; }
  const float shy = block_sum_256(ys, sh) / (float)HW;
  const float sht = block_sum_256(ts, sh) / (float)HW;
  float* sY = sm;
  float* sT = sm + HW;
  if (in_lds) {
    for (int e = tid; e < HW; e += 256) { sY[e] = yn[e] - shy; sT[e] = tn[e] - sht; }
  }
  __syncthreads();
  const int OH = (H - ws) / stride + 1, OW = (W - ws) / stride + 1;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float ss = 0.f;
  for (int e = tid; e < OH * OW; e += 256) {
    const int i0 = (e / OW) * stride, j0 = (e % OW) * stride;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
    for (int a = 0; a < ws; ++a) {
      float h1 = 0.f, h2 = 0.f, h11 = 0.f, h22 = 0.f, h12 = 0.f;
      for (int b = 0; b < ws; ++b) {
        const int q = (i0 + a) * W + j0 + b;
        const float u = in_lds ? sY[q] : yn[q] - shy, v = in_lds ? sT[q] : tn[q] - sht, w = g1[b];
        h1 += w * u; h2 += w * v; h11 += w * (u * u); h22 += w * (v * v); h12 += w * (u * v);
      }
      const float w = g1[a];
      m1 += w * h1; m2 += w * h2; e11 += w * h11; e22 += w * h22; e12 += w * h12;
    }
    const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    const float u1 = m1 + shy, u2 = m2 + sht;
    ss += ((2.f * u1 * u2 + C1) * (2.f * s12 + C2)) / ((u1 * u1 + u2 * u2 + C1) * (s11 + s22 + C2));
  }
  const float SS = block_sum_256(ss, sh);
  if (tid == 0) { sums[4 * n + 0] = 0.f; sums[4 * n + 1] = 0.f; sums[4 * n + 2] = SS; sums[4 * n + 3] = 0.f; }
}

void launch_ssim_general(const float* y, const float* t, int N, int H, int W, int ws, int stride, int uniform, float* sums,
                         hipStream_t s) {
  DBM_CHECK(ws >= 1 && ws <= 64 && stride >= 1, "ssim: window_size must be in [1, 64], stride >= 1");
  DBM_CHECK(H >= ws && W >= ws, "ssim: the images are smaller than the window");
  const size_t lds = 2 * sizeof(float) * (size_t)H * W;
  const int in_lds = lds <= 60 * 1024;
  hipLaunchKernelGGL(ssim_general_kernel, dim3(N), dim3(256), in_lds ? lds : 0, s, y, t, H, W, ws, stride, uniform, in_lds, sums);
  DBM_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------
// chainer.optimizers.Adam  (srgan_train.py:1043-1048; AdamRule.update_core)
// ----------------------------------------------------------------------------------------------
// The sticky timeout flag is sampled ONCE per optimizer launch (a persistent kernel on another stream may raise it while
// the update runs: per-block reads would leave a model half updated): gate[0] = flag, and a no-op launch is counted so
// that the host can take the step count back (adam_t).
__global__ void adam_gate_kernel(const int* flag, int* gate, int* skipped) {
  const int f = *flag;
  *gate = f;
  if (f) *skipped += 1;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float alpha_t,
                                                   float omb1, float omb2, float eps, float gscale, const int* gate) {
  if (gate && *gate) return;  // (wave-uniform scalar load; written by adam_gate_kernel just before this launch)
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float gr = g[e] * gscale;
    float mm = m[e], vv = v[e];
    mm += omb1 * (gr - mm);
    vv += omb2 * (gr * gr - vv);
    m[e] = mm;
    v[e] = vv;
    p[e] -= alpha_t * mm / (sqrtf(vv) + eps);
  }
}

void launch_adam(float* p, const float* g, float* m, float* v, long n, float alpha_t, float one_minus_beta1,
                 float one_minus_beta2, float eps, float gscale, hipStream_t s, const int* skip, int* skipped) {
  if (dbm_abl_skip() & 4) return;  // (libdbm_measure.so only)
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const int* gate = nullptr;
  if (skip && skipped) {  // skipped[0] = count of no-op launches, skipped[1] = this launch's gate word
    hipLaunchKernelGGL(adam_gate_kernel, dim3(1), dim3(1), 0, s, skip, skipped + 1, skipped);
    gate = skipped + 1;
  }
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, alpha_t, one_minus_beta1,
                     one_minus_beta2, eps, gscale, gate);
  DBM_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long n, float v) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) p[e] = v;
}

void launch_fill(float* p, long n, float v, hipStream_t s) {
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
  DBM_HIP(hipGetLastError());
}

// np.clip(a, a_min=lo, a_max=None) in place (deepbedmap.py:663-665: W1, W2, W3 clipped to >= 0 before the sweep); NaN stays NaN
__global__ __launch_bounds__(256) void clip_min_kernel(float* __restrict__ p, long n, float lo) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float v = p[e];
    if (v < lo) p[e] = lo;
  }
}

void launch_clip_min(float* p, long n, float lo, hipStream_t s) {
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(clip_min_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, lo);
  DBM_HIP(hipGetLastError());
}

// sum of squared differences (psnr, srgan_train.py:906-928)
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                     float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float d = a[e] - b[e];
    s += d * d;
  }
  const float t = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = t;  // per-block partial sums, added in order by the caller's finishing kernel
}

int sqdiff_blocks(long n) {
  long blocks = (n + 255) / 256;
  return (int)(blocks > 1024 ? 1024 : blocks);
}

void launch_sqdiff(const float* a, const float* b, long n, float* out, hipStream_t s) {
  const long blocks = sqdiff_blocks(n);
  hipLaunchKernelGGL(sqdiff_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, n, out);
  DBM_HIP(hipGetLastError());
}
