"""Independent torch-CPU (autograd) restatement of the hot path -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Two uses: cross-checking the NumPy oracle's forward and hand-written backward in float64
(tests/test_oracle_vs_torch.py), and the "torch_cpu" figure of bench.py's cpu_baseline leg (the same training
iteration in float32 through oneDNN: a strong-CPU yardstick next to the NumPy/BLAS port).
It is written from the same semantics sheet but shares no code with the NumPy modules:
convs go through torch.nn.functional (oneDNN), gradients through autograd."""
import math

import torch
import torch.nn.functional as F

SLOPE = 0.2


def tp(params, dtype=torch.float64):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params.items()}


def conv(P, name, x, stride=1, pad=1, bias=True):
    return F.conv2d(x, P[name + "/W"], P[name + "/b"] if bias else None, stride=stride, padding=pad)


def deform_conv(x, offset, W, b, pad=1):
    """Chainer deformable_convolution_2d_sampler semantics, stride 1."""
    N, C, H, Wd = x.shape
    O, _, kh, kw = W.shape
    kk = kh * kw
    OH, OW = offset.shape[2:]
    Hp, Wp = H + 2 * pad, Wd + 2 * pad
    ys = torch.arange(OH, dtype=x.dtype).view(1, 1, OH, 1)
    xs = torch.arange(OW, dtype=x.dtype).view(1, 1, 1, OW)
    fx = torch.arange(kw, dtype=x.dtype).repeat(kh).view(1, kk, 1, 1)
    fy = torch.arange(kh, dtype=x.dtype).repeat_interleave(kw).view(1, kk, 1, 1)
    xc = offset[:, :kk] + xs + fx  # position in the conv-padded frame
    yc = offset[:, kk:] + ys + fy
    u = xc + 1  # position in the sampler's doubly padded frame
    v = yc + 1
    uc = u.clamp(0, Wp + 1)
    vc = v.clamp(0, Hp + 1)
    u0 = uc.detach().floor().clamp(0, Wp)
    v0 = vc.detach().floor().clamp(0, Hp)
    xpp = F.pad(x, (pad + 1, pad + 1, pad + 1, pad + 1))
    Wpp = Wp + 2
    flat = xpp.reshape(N, C, -1)

    def gather(vi, ui):
        idx = (vi * Wpp + ui).long().reshape(N, 1, -1).expand(N, C, -1)
        return torch.gather(flat, 2, idx).reshape(N, C, kk, OH, OW)

    wu0 = (uc - u0).unsqueeze(1)
    wu1 = (u0 + 1 - uc).unsqueeze(1)
    wv0 = (vc - v0).unsqueeze(1)
    wv1 = (v0 + 1 - vc).unsqueeze(1)
    col = (wu1 * wv1) * gather(v0, u0) + (wu0 * wv1) * gather(v0, u0 + 1) \
        + (wu1 * wv0) * gather(v0 + 1, u0) + (wu0 * wv0) * gather(v0 + 1, u0 + 1)
    y = torch.einsum("nckhw,ock->nohw", col, W.reshape(O, C, kk))
    return y + b.view(1, -1, 1, 1)


def generator_forward(P, x, w1, w2, w3, n_blocks, rs):
    a0 = torch.cat([
        conv(P, "input_block/conv_on_X", x, 1, 0),
        conv(P, "input_block/conv_on_W1", w1, 10, 0),
        conv(P, "input_block/conv_on_W2", w2, 2, 0),
        conv(P, "input_block/conv_on_W3", w3, 1, 0),
    ], dim=1)
    a1 = F.leaky_relu(conv(P, "pre_residual_conv_layer", a0), SLOPE)
    h = a1
    for i in range(n_blocks):
        xin = h
        for d in (1, 2, 3):
            base = f"residual_network/{i}/residual_dense_block{d}"
            cat = h
            for k in (1, 2, 3, 4):
                cat = torch.cat([cat, F.leaky_relu(conv(P, f"{base}/conv_layer{k}", cat), SLOPE)], dim=1)
            h = conv(P, f"{base}/conv_layer5", cat) * rs + h
        h = h * rs + xin
    a3 = a1 + conv(P, "post_residual_conv_layer", h)
    a41 = F.leaky_relu(conv(P, "post_upsample_conv_layer_1", F.interpolate(a3, scale_factor=2, mode="nearest")), SLOPE)
    a42 = F.leaky_relu(conv(P, "post_upsample_conv_layer_2", F.interpolate(a41, scale_factor=2, mode="nearest")), SLOPE)
    off1 = conv(P, "final_conv_layer1/offset_conv", a42)
    a51 = F.leaky_relu(deform_conv(a42, off1, P["final_conv_layer1/deform_conv/W"], P["final_conv_layer1/deform_conv/b"]), SLOPE)
    off2 = conv(P, "final_conv_layer2/offset_conv", a51)
    return deform_conv(a51, off2, P["final_conv_layer2/deform_conv/W"], P["final_conv_layer2/deform_conv/b"])


D_STRIDES = [1, 2, 1, 2, 1, 2, 1, 2, 1, 2]


def discriminator_forward(P, S, x, train=True):
    """S: dict of running stats tensors (updated in place when train)."""
    h = F.leaky_relu(conv(P, "conv_layer0", x, 1, 1), SLOPE)
    for i in range(1, 10):
        z = F.conv2d(h, P[f"conv_layer{i}/W"], None, stride=D_STRIDES[i], padding=1)
        g, b = P[f"batch_norm{i}/gamma"], P[f"batch_norm{i}/beta"]
        if train:
            m = z.shape[0] * z.shape[2] * z.shape[3]
            mean = z.mean(dim=(0, 2, 3))
            var = z.var(dim=(0, 2, 3), unbiased=False)
            zn = (z - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
            with torch.no_grad():
                S[f"batch_norm{i}/avg_mean"].mul_(0.9).add_(0.1 * mean)
                S[f"batch_norm{i}/avg_var"].mul_(0.9).add_(0.1 * var * (m / max(m - 1.0, 1.0)))
        else:
            zn = (z - S[f"batch_norm{i}/avg_mean"].view(1, -1, 1, 1)) / torch.sqrt(
                S[f"batch_norm{i}/avg_var"].view(1, -1, 1, 1) + 1e-5)
        h = F.leaky_relu(zn * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1), SLOPE)
    flat = h.reshape(len(h), -1)
    l1 = F.leaky_relu(flat @ P["linear_1/W"].t() + P["linear_1/b"], SLOPE)
    return l1 @ P["linear_2/W"].t() + P["linear_2/b"]


def sce(x, t):
    return F.binary_cross_entropy_with_logits(x, t.to(x.dtype), reduction="mean")


def d_loss(real, fake):
    return sce(real - fake.mean(), torch.ones_like(real)) + sce(fake - real.mean(), torch.zeros_like(fake))


def ssim(y, t, window_size=9, kind="gaussian"):
    if kind == "uniform":
        g = torch.ones(window_size, dtype=y.dtype)
    else:
        g = torch.tensor([math.exp(-((i - window_size // 2) ** 2) / (2 * 1.5 ** 2)) for i in range(window_size)], dtype=y.dtype)
    g = g / g.sum()
    win = (g[:, None] * g[None, :]).view(1, 1, window_size, window_size)
    mu1, mu2 = F.conv2d(y, win), F.conv2d(t, win)
    s11 = F.conv2d(y * y, win) - mu1 * mu1
    s22 = F.conv2d(t * t, win) - mu2 * mu2
    s12 = F.conv2d(y * t, win) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))).mean()


def g_loss(y_pred, y_true, fake_labels, x_topo, kind="gaussian"):
    real_labels = torch.ones_like(fake_labels)
    adv = sce(real_labels - fake_labels.mean(), torch.zeros_like(real_labels)) + sce(
        fake_labels - real_labels.mean(), torch.ones_like(fake_labels))
    content = (y_pred - y_true).abs().mean()
    topo = (F.avg_pool2d(y_pred, 4) - x_topo).abs().mean()
    return 1e-2 * content + 2e-2 * adv + 2e-3 * topo + 5.25 * (1 - ssim(y_pred, y_true, 9, kind))


def training_iteration(Pg, Pd, Sd, arrays, n_blocks=12, rs=0.1, lr=1.6e-4):
    """One D-step + G-step (srgan_train.py:1084-1263) through autograd: what bench.py times on the host cores.  The
    update is a sign step (what Adam's first step amounts to); the optimizer is a negligible part of the iteration."""
    X, W1, W2, W3, Y = (arrays[k] for k in ("X", "W1", "W2", "W3", "Y"))
    for p in list(Pg.values()) + list(Pd.values()):
        p.grad = None
    # D-step (:1131-1164): fakes under no_grad, two training-mode BatchNorm batches, RaGAN loss, backward
    with torch.no_grad():
        fake = generator_forward(Pg, X, W1, W2, W3, n_blocks, rs)
    real_pred = discriminator_forward(Pd, Sd, Y, train=True)
    fake_pred = discriminator_forward(Pd, Sd, fake, train=True)
    dl = d_loss(real_pred, fake_pred)
    dl.backward()
    with torch.no_grad():
        for p in Pd.values():
            p -= lr * torch.sign(p.grad)
    # G-step (:1222-1257): forward with graph, eval-mode detached discriminator, four-term loss, backward
    for p in Pg.values():
        p.grad = None
    fake = generator_forward(Pg, X, W1, W2, W3, n_blocks, rs)
    with torch.no_grad():
        fake_labels = discriminator_forward(Pd, Sd, fake, train=False)
    gl = g_loss(fake, Y, fake_labels, X[:, :, 1:-1, 1:-1])
    gl.backward()
    with torch.no_grad():
        for p in Pg.values():
            p -= lr * torch.sign(p.grad)
    return float(dl.detach()), float(gl.detach())
