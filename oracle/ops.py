"""NumPy restatement of the Chainer 7.0.0 functions used on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function cites the reference call
site (file:line under /root/reference) it stands in for and the upstream Chainer
function whose published algorithm it restates.  Forward AND backward are explicit so
that the HIP kernels' gradients can be checked.

All arithmetic runs in the dtype of the inputs (float32 for the model; the reference's
loss doctests feed float64, which is honoured).
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

LRELU_SLOPE = 0.2  # srgan_train.py:340 (and every other F.leaky_relu call site)


# --------------------------------------------------------------------------------------
# L.Convolution2D  (chainer/functions/connection/convolution_2d.py: im2col_cpu + tensordot)
# call sites: srgan_train.py:223-254, 292-331, 467-503, 617-634
# --------------------------------------------------------------------------------------
def _im2col(x, kh, kw, s, p):
    """(N,C,H,W) -> (N,C,OH,OW,kh,kw) view on the zero-padded input (cover_all=False)."""
    if p:
        x = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    win = sliding_window_view(x, (kh, kw), axis=(2, 3))
    return win[:, :, ::s, ::s]


def conv2d(x, W, b=None, stride=1, pad=0):
    """Cross-correlation, OIHW weights, zero padding, out = floor((H+2p-k)/s)+1."""
    O, C, kh, kw = W.shape
    col = _im2col(x, kh, kw, stride, pad)  # N,C,OH,OW,kh,kw
    N, _, OH, OW = col.shape[:4]
    colm = np.ascontiguousarray(col.transpose(0, 2, 3, 1, 4, 5)).reshape(N * OH * OW, C * kh * kw)
    y = colm @ W.reshape(O, -1).T
    if b is not None:
        y = y + b
    return np.ascontiguousarray(y.reshape(N, OH, OW, O).transpose(0, 3, 1, 2))


def conv2d_backward(x, W, gy, stride=1, pad=0, need_gx=True):
    """Returns (gx, gW, gb).  gW = gy (x) im2col(x); gx = col2im(W^T gy)."""
    O, C, kh, kw = W.shape
    N, _, H, Wd = x.shape
    col = _im2col(x, kh, kw, stride, pad)
    OH, OW = col.shape[2:4]
    colm = np.ascontiguousarray(col.transpose(0, 2, 3, 1, 4, 5)).reshape(N * OH * OW, C * kh * kw)
    gym = np.ascontiguousarray(gy.transpose(0, 2, 3, 1)).reshape(N * OH * OW, O)
    gW = (gym.T @ colm).reshape(W.shape)
    gb = gym.sum(axis=0)
    gx = None
    if need_gx:
        gcol = (gym @ W.reshape(O, -1)).reshape(N, OH, OW, C, kh, kw)
        gxp = np.zeros((N, C, H + 2 * pad, Wd + 2 * pad), dtype=x.dtype)
        for ky in range(kh):
            for kx in range(kw):
                gxp[:, :, ky:ky + stride * OH:stride, kx:kx + stride * OW:stride] += gcol[
                    :, :, :, :, ky, kx
                ].transpose(0, 3, 1, 2)
        gx = gxp[:, :, pad:pad + H, pad:pad + Wd] if pad else gxp
    return gx, gW, gb


# --------------------------------------------------------------------------------------
# F.leaky_relu(slope=0.2)  (chainer/functions/activation/leaky_relu.py)
# --------------------------------------------------------------------------------------
def leaky_relu(x, slope=LRELU_SLOPE):
    return np.where(x >= 0, x, x * x.dtype.type(slope))


def leaky_relu_backward(y, gy, slope=LRELU_SLOPE):
    """Chainer retains the OUTPUT and tests y >= 0 (same sign as x for slope > 0)."""
    return np.where(y >= 0, gy, gy * gy.dtype.type(slope))


# --------------------------------------------------------------------------------------
# F.resize_images(mode="nearest") x2   srgan_train.py:556-558, 562-566
# out[i, j] = in[floor(i*H/H_out), floor(j*W/W_out)] = in[i//2, j//2] for an exact x2.
# --------------------------------------------------------------------------------------
def upsample_nearest2(x):
    return x.repeat(2, axis=2).repeat(2, axis=3)


def upsample_nearest2_backward(gy):
    N, C, H2, W2 = gy.shape
    return gy.reshape(N, C, H2 // 2, 2, W2 // 2, 2).sum(axis=(3, 5))


# --------------------------------------------------------------------------------------
# L.DeformableConvolution2D  srgan_train.py:506-523, forward at 572-574
# (chainer/links/connection/deformable_convolution_2d.py,
#  chainer/functions/connection/deformable_convolution_2d_sampler.py,
#  chainer/functions/array/spatial_transformer_sampler.py)
# --------------------------------------------------------------------------------------
def _deform_geometry(offset, H, W, kh, kw, stride, pad):
    """Sampling corners / weights exactly as _offset2grid + spatial_transformer_sampler.

    offset: (N, 2*kh*kw, OH, OW); channels [0:khkw] are x offsets, [khkw:] y offsets.
    Returns u0,v0 (int, in the doubly padded frame), fractional parts and clip masks,
    each shaped (N, khkw, OH*OW).
    """
    N, _, OH, OW = offset.shape
    kk = kh * kw
    f = offset.dtype.type
    Hp, Wp = H + 2 * pad, W + 2 * pad  # size of the conv-padded image fed to the sampler
    ys, xs = np.meshgrid(
        np.arange(0, stride * OH, stride, dtype=np.float32),
        np.arange(0, stride * OW, stride, dtype=np.float32),
        indexing="ij",
    )
    fx = np.tile(np.arange(kw, dtype=np.float32), kh)
    fy = np.repeat(np.arange(kh, dtype=np.float32), kw)
    x_coord = offset[:, :kk] + xs[None, None] + fx[None, :, None, None]
    y_coord = offset[:, kk:] + ys[None, None] + fy[None, :, None, None]
    # normalise to [-1, 1] ... (deformable_convolution_2d_sampler._offset2grid)
    x_coord = (x_coord / f(Wp - 1) - f(0.5)) * f(2)
    y_coord = (y_coord / f(Hp - 1) - f(0.5)) * f(2)
    u = x_coord.reshape(N, kk, OH * OW)
    v = y_coord.reshape(N, kk, OH * OW)
    # ... and back, shifted by the sampler's own 1-pixel zero ring (spatial_transformer_sampler)
    u = (u + f(1)) * f(Wp - 1) / f(2) + f(1)
    v = (v + f(1)) * f(Hp - 1) / f(2) + f(1)
    uc = np.clip(u, 0, Wp + 1)
    vc = np.clip(v, 0, Hp + 1)
    u0 = np.clip(np.floor(uc).astype(np.int32), 0, Wp)
    v0 = np.clip(np.floor(vc).astype(np.int32), 0, Hp)
    return u, v, uc, vc, u0, v0, Hp, Wp


def deform_conv2d(x, offset, W, b=None, stride=1, pad=1, return_cache=False):
    """F.deformable_convolution_2d_sampler: bilinear-sample the zero padded input at
    (regular tap position + learned offset), then GEMM with W (O,C,kh,kw)."""
    N, C, H, Wd = x.shape
    O, _, kh, kw = W.shape
    kk = kh * kw
    OH, OW = offset.shape[2:]
    u, v, uc, vc, u0, v0, Hp, Wp = _deform_geometry(offset, H, Wd, kh, kw, stride, pad)
    u1, v1 = u0 + 1, v0 + 1
    xpp = np.pad(x, ((0, 0), (0, 0), (pad + 1, pad + 1), (pad + 1, pad + 1)))
    wu0 = (uc - u0).astype(x.dtype)
    wu1 = (u1 - uc).astype(x.dtype)
    wv0 = (vc - v0).astype(x.dtype)
    wv1 = (v1 - vc).astype(x.dtype)
    col = np.empty((N, C, kk, OH * OW), dtype=x.dtype)
    for n in range(N):
        x1 = xpp[n][:, v0[n], u0[n]]  # C,kk,P
        x2 = xpp[n][:, v0[n], u1[n]]
        x3 = xpp[n][:, v1[n], u0[n]]
        x4 = xpp[n][:, v1[n], u1[n]]
        col[n] = (wu1[n] * wv1[n]) * x1 + (wu0[n] * wv1[n]) * x2 + (wu1[n] * wv0[n]) * x3 + (wu0[n] * wv0[n]) * x4
    colm = np.ascontiguousarray(col.transpose(0, 3, 1, 2)).reshape(N * OH * OW, C * kk)
    y = colm @ W.reshape(O, -1).T
    if b is not None:
        y = y + b
    y = np.ascontiguousarray(y.reshape(N, OH, OW, O).transpose(0, 3, 1, 2))
    if return_cache:
        return y, (colm,)
    return y


def deform_conv2d_backward(x, offset, W, gy, stride=1, pad=1):
    """Returns (gx, goffset, gW, gb).  Gradients of the bilinear sampler follow
    SpatialTransformerSampler._backward (coordinate gradient masked where clipped)."""
    N, C, H, Wd = x.shape
    O, _, kh, kw = W.shape
    kk = kh * kw
    OH, OW = offset.shape[2:]
    P = OH * OW
    u, v, uc, vc, u0, v0, Hp, Wp = _deform_geometry(offset, H, Wd, kh, kw, stride, pad)
    u1, v1 = u0 + 1, v0 + 1
    xpp = np.pad(x, ((0, 0), (0, 0), (pad + 1, pad + 1), (pad + 1, pad + 1)))
    wu0 = (uc - u0).astype(x.dtype)
    wu1 = (u1 - uc).astype(x.dtype)
    wv0 = (vc - v0).astype(x.dtype)
    wv1 = (v1 - vc).astype(x.dtype)
    gym = np.ascontiguousarray(gy.transpose(0, 2, 3, 1)).reshape(N * P, O)
    gb = gym.sum(axis=0)
    # recompute col for gW
    col = np.empty((N, C, kk, P), dtype=x.dtype)
    gcol = (gym @ W.reshape(O, -1)).reshape(N, P, C, kk).transpose(0, 2, 3, 1)  # N,C,kk,P
    gxpp = np.zeros_like(xpp)
    gu = np.empty((N, kk, P), dtype=x.dtype)
    gv = np.empty((N, kk, P), dtype=x.dtype)
    for n in range(N):
        x1 = xpp[n][:, v0[n], u0[n]]
        x2 = xpp[n][:, v0[n], u1[n]]
        x3 = xpp[n][:, v1[n], u0[n]]
        x4 = xpp[n][:, v1[n], u1[n]]
        col[n] = (wu1[n] * wv1[n]) * x1 + (wu0[n] * wv1[n]) * x2 + (wu1[n] * wv0[n]) * x3 + (wu0[n] * wv0[n]) * x4
        g = gcol[n]
        gu[n] = (g * (-wv1[n] * x1 + wv1[n] * x2 - wv0[n] * x3 + wv0[n] * x4)).sum(axis=0)
        gv[n] = (g * (-wu1[n] * x1 - wu0[n] * x2 + wu1[n] * x3 + wu0[n] * x4)).sum(axis=0)
        cidx = np.arange(C)[:, None, None]
        np.add.at(gxpp[n], (cidx, v0[n][None], u0[n][None]), g * (wu1[n] * wv1[n]))
        np.add.at(gxpp[n], (cidx, v0[n][None], u1[n][None]), g * (wu0[n] * wv1[n]))
        np.add.at(gxpp[n], (cidx, v1[n][None], u0[n][None]), g * (wu1[n] * wv0[n]))
        np.add.at(gxpp[n], (cidx, v1[n][None], u1[n][None]), g * (wu0[n] * wv0[n]))
    colm = np.ascontiguousarray(col.transpose(0, 3, 1, 2)).reshape(N * P, C * kk)
    gW = (gym.T @ colm).reshape(W.shape)
    # coordinate gradient: (W-1)/2 from the sampler times 2/(W-1) from _offset2grid = 1,
    # zeroed where the coordinate was clipped (u <= 0 or u >= W+1 in the doubly padded frame).
    gu = gu * ((u > 0) & (u < Wp + 1))
    gv = gv * ((v > 0) & (v < Hp + 1))
    goffset = np.concatenate([gu, gv], axis=1).reshape(N, 2 * kk, OH, OW).astype(x.dtype)
    q = pad + 1
    gx = gxpp[:, :, q:q + H, q:q + Wd]
    return np.ascontiguousarray(gx), goffset, gW, gb


# --------------------------------------------------------------------------------------
# L.BatchNormalization(axis=(0,2,3), eps=1e-5)  srgan_train.py:636-644
# (chainer/functions/normalization/batch_normalization.py; link decay = 0.9)
# --------------------------------------------------------------------------------------
def batchnorm_train(x, gamma, beta, avg_mean, avg_var, eps=1e-5, decay=0.9):
    """Batch statistics (biased var); running stats updated in place with the unbiased
    correction m/max(m-1,1).  Returns y and the cache needed for backward."""
    f = x.dtype.type
    m = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(axis=(0, 2, 3))
    var = x.var(axis=(0, 2, 3))
    inv_std = f(1) / np.sqrt(var + f(eps))
    xhat = (x - mean[None, :, None, None]) * inv_std[None, :, None, None]
    y = gamma[None, :, None, None] * xhat + beta[None, :, None, None]
    adjust = m / max(m - 1.0, 1.0)
    avg_mean *= f(decay)
    avg_mean += f(1 - decay) * mean
    avg_var *= f(decay)
    avg_var += f((1 - decay) * adjust) * var
    return y, (xhat, inv_std)


def batchnorm_eval(x, gamma, beta, avg_mean, avg_var, eps=1e-5):
    """chainer.config.train == False: fixed_batch_normalization with the running stats."""
    f = x.dtype.type
    inv_std = f(1) / np.sqrt(avg_var + f(eps))
    return (
        gamma[None, :, None, None] * (x - avg_mean[None, :, None, None]) * inv_std[None, :, None, None]
        + beta[None, :, None, None]
    )


def batchnorm_train_backward(gy, gamma, cache):
    xhat, inv_std = cache
    m = gy.shape[0] * gy.shape[2] * gy.shape[3]
    gbeta = gy.sum(axis=(0, 2, 3))
    ggamma = (gy * xhat).sum(axis=(0, 2, 3))
    f = gy.dtype.type
    gx = (gamma * inv_std)[None, :, None, None] * (
        gy - (gbeta[None, :, None, None] + xhat * ggamma[None, :, None, None]) / f(m)
    )
    return gx, ggamma, gbeta


# --------------------------------------------------------------------------------------
# L.Linear  srgan_train.py:646-647, 693-696   y = x W^T + b, W (out,in)
# --------------------------------------------------------------------------------------
def linear(x, W, b):
    return x @ W.T + b


def linear_backward(x, W, gy):
    return gy @ W, gy.T @ x, gy.sum(axis=0)


# --------------------------------------------------------------------------------------
# Losses / metrics
# --------------------------------------------------------------------------------------
def mean_absolute_error(a, b):
    """F.mean_absolute_error  srgan_train.py:871, 882: sum|a-b| / a.size."""
    d = a - b
    return np.abs(d).sum() / d.dtype.type(d.size)


def mean_absolute_error_backward(a, b):
    """d/da; Chainer uses sign(diff)/size (sign(0) = 0)."""
    d = a - b
    return np.sign(d) / d.dtype.type(d.size)


def average_pooling_4x4(x):
    """F.average_pooling_2d(ksize=(4,4))  srgan_train.py:883 (stride = ksize, pad 0)."""
    N, C, H, W = x.shape
    return x.reshape(N, C, H // 4, 4, W // 4, 4).mean(axis=(3, 5))


def average_pooling_4x4_backward(gy):
    return (gy / gy.dtype.type(16)).repeat(4, axis=2).repeat(4, axis=3)


def sigmoid_cross_entropy(x, t):
    """F.sigmoid_cross_entropy (normalize=True, reduce='mean')  srgan_train.py:999-1004.
    loss = sum(-(x*(t-[x>=0]) - log1p(exp(-|x|)))) / max(count(t != -1), 1)"""
    ignore = t != -1
    loss = -(ignore * (x * (t - (x >= 0)) - np.log1p(np.exp(-np.abs(x)))))
    count = max(int(ignore.sum()), 1)
    return loss.sum() / x.dtype.type(count)


def sigmoid_cross_entropy_backward(x, t):
    ignore = t != -1
    count = max(int(ignore.sum()), 1)
    sig = 1.0 / (1.0 + np.exp(-x))
    return (ignore * (sig - t) / count).astype(x.dtype)


def binary_accuracy(y, t):
    """F.binary_accuracy  srgan_train.py:1158: mean((y >= 0) == t) over t != -1."""
    ignore = t != -1
    pred = y >= 0
    return ((pred == t) & ignore).sum() / max(int(ignore.sum()), 1)


def psnr(y_pred, y_true, data_range=2 ** 32):
    """srgan_train.py:906-928 verbatim semantics."""
    mse = np.mean(np.square(np.subtract(y_pred, y_true)), axis=None)
    return np.multiply(20, np.log10(data_range / np.sqrt(mse)))


# ssim.functions.ssim_loss(y, t, window_size=9, stride=1)  srgan_train.py:953
# (ssim-chainer @ 9c54f25; valid windows, normalised window weights, C1/C2 from :833)
SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


def ssim_window(window_size=9, kind="gaussian", sigma=1.5, dtype=np.float32):
    if kind == "uniform":
        g = np.ones(window_size, dtype=np.float64)
    else:
        c = window_size // 2
        g = np.exp(-((np.arange(window_size) - c) ** 2) / (2.0 * sigma ** 2))
    g = g / g.sum()
    return np.outer(g, g).astype(dtype)


def _win_filter(img, win, stride):
    k = win.shape[0]
    v = sliding_window_view(img, (k, k), axis=(2, 3))[:, :, ::stride, ::stride]
    return np.tensordot(v, win, axes=((4, 5), (0, 1)))


def _ssim_terms(y, t, win, stride):
    f = y.dtype.type
    mu1 = _win_filter(y, win, stride)
    mu2 = _win_filter(t, win, stride)
    e11 = _win_filter(y * y, win, stride)
    e22 = _win_filter(t * t, win, stride)
    e12 = _win_filter(y * t, win, stride)
    s11 = e11 - mu1 * mu1
    s22 = e22 - mu2 * mu2
    s12 = e12 - mu1 * mu2
    A1 = f(2) * mu1 * mu2 + f(SSIM_C1)
    A2 = f(2) * s12 + f(SSIM_C2)
    B1 = mu1 * mu1 + mu2 * mu2 + f(SSIM_C1)
    B2 = s11 + s22 + f(SSIM_C2)
    return mu1, mu2, A1, A2, B1, B2


def ssim(y, t, window_size=9, stride=1, kind="gaussian"):
    if y.shape != t.shape:
        raise ValueError("Input images must have the same dimensions.")  # srgan_train.py:950-951
    win = ssim_window(window_size, kind, dtype=y.dtype)
    _, _, A1, A2, B1, B2 = _ssim_terms(y, t, win, stride)
    return ((A1 * A2) / (B1 * B2)).mean()


def ssim_backward(y, t, window_size=9, stride=1, kind="gaussian"):
    """d mean(SSIM) / d y (t is a constant)."""
    f = y.dtype.type
    win = ssim_window(window_size, kind, dtype=y.dtype)
    mu1, mu2, A1, A2, B1, B2 = _ssim_terms(y, t, win, stride)
    smap = (A1 * A2) / (B1 * B2)
    cnt = f(smap.size)
    g_mu1 = smap * (f(2) * mu2 / A1 - f(2) * mu2 / A2 - f(2) * mu1 / B1 + f(2) * mu1 / B2) / cnt
    g_e11 = -smap / B2 / cnt
    g_e12 = f(2) * smap / A2 / cnt
    k = window_size

    def scatter(g):
        out = np.zeros_like(y)
        OH, OW = g.shape[2:]
        for i in range(k):
            for j in range(k):
                out[:, :, i:i + stride * OH:stride, j:j + stride * OW:stride] += g * win[i, j]
        return out

    return scatter(g_mu1) + f(2) * y * scatter(g_e11) + t * scatter(g_e12)


# --------------------------------------------------------------------------------------
# chainer.optimizers.Adam  srgan_train.py:1043-1048 (chainer/optimizers/adam.py, AdamRule)
# --------------------------------------------------------------------------------------
def adam_update(p, g, m, v, t, alpha, beta1=0.9, beta2=0.999, eps=1e-8, eta=1.0, wd=0.0):
    """In place; t is the 1-based step count AFTER increment."""
    f = p.dtype.type
    m += f(1 - beta1) * (g - m)
    v += f(1 - beta2) * (g * g - v)
    fix1 = 1.0 - beta1 ** t
    fix2 = 1.0 - beta2 ** t
    alpha_t = alpha * np.sqrt(fix2) / fix1
    p -= f(eta) * (f(alpha_t) * m / (np.sqrt(v) + f(eps)) + f(wd) * p)
