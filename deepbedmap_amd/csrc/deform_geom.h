// Sampling geometry of the deformable convolution (reference srgan_train.py:506-523, :572-574; Chainer
// deformable_convolution_2d_sampler + spatial_transformer_sampler semantics), shared by the sampler
// kernels of misc.hip and the fused sampler + GEMM kernels of deform_fused.hip.
#pragma once
#include <hip/hip_runtime.h>

struct DeformGeom {
  int u0, v0;            // top-left corner in the sampler's doubly padded frame
  float wu0, wu1, wv0, wv1;
  bool mu, mv;           // coordinate-gradient masks (not clipped)
};

__device__ __forceinline__ DeformGeom deform_geom(float offx, float offy, int a, int b, int ky, int kx, int H, int W,
                                                  int pad) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  // _offset2grid: normalise to [-1,1] in fp32, then spatial_transformer_sampler maps back (+1 for its zero ring)
  float xc = offx + (float)b + (float)kx;
  float yc = offy + (float)a + (float)ky;
  xc = (xc / (float)(Wp - 1) - 0.5f) * 2.f;
  yc = (yc / (float)(Hp - 1) - 0.5f) * 2.f;
  const float u = (xc + 1.f) * (float)(Wp - 1) / 2.f + 1.f;
  const float v = (yc + 1.f) * (float)(Hp - 1) / 2.f + 1.f;
  const float uc = fminf(fmaxf(u, 0.f), (float)(Wp + 1));
  const float vc = fminf(fmaxf(v, 0.f), (float)(Hp + 1));
  DeformGeom g;
  g.u0 = min(max((int)floorf(uc), 0), Wp);
  g.v0 = min(max((int)floorf(vc), 0), Hp);
  g.wu0 = uc - (float)g.u0;
  g.wu1 = (float)(g.u0 + 1) - uc;
  g.wv0 = vc - (float)g.v0;
  g.wv1 = (float)(g.v0 + 1) - vc;
  g.mu = (u > 0.f) && (u < (float)(Wp + 1));
  g.mv = (v > 0.f) && (v < (float)(Hp + 1));
  return g;
}

// corner (vv,uu) of the doubly padded frame -> offset into the unpadded image or -1
__device__ __forceinline__ int deform_corner(int vv, int uu, int H, int W, int pad) {
  const int y = vv - pad - 1, x = uu - pad - 1;
  return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? y * W + x : -1;
}

