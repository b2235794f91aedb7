"""Cross-check of the NumPy oracle (forward + hand-written backward) against an independent
torch-CPU autograd restatement, in float64 so that any disagreement is a semantic one."""
import numpy as np
import pytest
import torch

from oracle import model, ops, train
from oracle import torch_ref as tr

torch.set_num_threads(4)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _inputs(n, rs, dtype=np.float64, h=11, w=11):
    return (rs.rand(n, 1, h, w).astype(dtype), rs.rand(n, 1, 10 * h, 10 * w).astype(dtype),
            rs.rand(n, 2, 2 * h, 2 * w).astype(dtype), rs.rand(n, 1, h, w).astype(dtype))


def _scaled_generator(n_blocks, scale, dtype=np.float64, seed=3):
    g = model.GeneratorModel(num_residual_blocks=n_blocks, residual_scaling=0.3, seed=seed, dtype=dtype)
    rs = np.random.RandomState(seed + 1)
    for k in g.params:
        if k.endswith("/W"):
            g.params[k] *= scale
        else:
            g.params[k] += rs.normal(0, 0.1, g.params[k].shape)  # non-zero biases
    return g


@pytest.mark.parametrize("scale", [1.0, 10.0])
def test_generator_forward_backward_vs_torch(scale):
    rs = np.random.RandomState(7)
    g = _scaled_generator(2, scale)
    x, w1, w2, w3 = _inputs(2, rs)
    y = g.forward(x, w1, w2, w3, keep=True)
    P = tr.tp(g.params)
    yt = tr.generator_forward(P, *(torch.tensor(a) for a in (x, w1, w2, w3)), 2, 0.3)
    assert rel(y, yt.detach().numpy()) < 1e-9
    gy = rs.normal(size=y.shape)
    (yt * torch.tensor(gy)).sum().backward()
    G = g.backward(gy)
    assert set(G) == set(g.params)
    for k in G:
        assert rel(G[k], P[k].grad.numpy()) < 1e-7, k


def test_deform_conv_large_offsets_and_border_clipping():
    """Offsets large enough to leave the image exercise the clip masks of the sampler."""
    rs = np.random.RandomState(11)
    x = rs.normal(size=(2, 5, 7, 6))
    off = rs.normal(scale=3.0, size=(2, 18, 7, 6))
    W = rs.normal(size=(4, 5, 3, 3))
    b = rs.normal(size=(4,))
    y = ops.deform_conv2d(x, off, W, b)
    xt, offt, Wt, bt = (torch.tensor(a, requires_grad=True) for a in (x, off, W, b))
    yt = tr.deform_conv(xt, offt, Wt, bt)
    assert rel(y, yt.detach().numpy()) < 1e-6  # oracle round-trips coordinates through float32 constants
    gy = rs.normal(size=y.shape)
    (yt * torch.tensor(gy)).sum().backward()
    gx, goff, gW, gb = ops.deform_conv2d_backward(x, off, W, gy)
    assert rel(gx, xt.grad.numpy()) < 1e-6
    assert rel(goff, offt.grad.numpy()) < 1e-5
    assert rel(gW, Wt.grad.numpy()) < 1e-6
    assert rel(gb, bt.grad.numpy()) < 1e-9


def test_deform_conv_zero_offset_is_regular_conv():  # SURVEY A.6 identity
    rs = np.random.RandomState(5)
    x = rs.normal(size=(1, 3, 8, 9)).astype(np.float32)
    W = rs.normal(size=(2, 3, 3, 3)).astype(np.float32)
    b = rs.normal(size=(2,)).astype(np.float32)
    y0 = ops.conv2d(x, W, b, 1, 1)
    y1 = ops.deform_conv2d(x, np.zeros((1, 18, 8, 9), np.float32), W, b)
    assert rel(y1, y0) < 1e-5


def test_discriminator_forward_backward_vs_torch():
    rs = np.random.RandomState(9)
    d = model.DiscriminatorModel(dtype=np.float64)
    for k in d.params:
        if k.endswith("/W"):
            d.params[k] *= 10.0
        elif k.endswith("gamma"):
            d.params[k] += rs.normal(0, 0.2, d.params[k].shape)
        else:
            d.params[k] += rs.normal(0, 0.1, d.params[k].shape)
    real = rs.rand(3, 1, 36, 36)
    fake = rs.rand(3, 1, 36, 36)
    P = tr.tp(d.params)
    S = {k: torch.tensor(v, dtype=torch.float64) for k, v in d.persistent.items() if not k.endswith("/N")}
    lr, cr = d.forward(real, train=True, keep=True)
    lf, cf = d.forward(fake, train=True, keep=True)
    lrt = tr.discriminator_forward(P, S, torch.tensor(real), True)
    lft = tr.discriminator_forward(P, S, torch.tensor(fake), True)
    assert rel(lr, lrt.detach().numpy()) < 1e-9 and rel(lf, lft.detach().numpy()) < 1e-9
    for k, v in S.items():
        assert rel(d.persistent[k], v.numpy()) < 1e-12, k
    t1, t0 = np.ones((3, 1), np.int32), np.zeros((3, 1), np.int32)
    loss = train.calculate_discriminator_loss(lr, lf, t1, t0)
    losst = tr.d_loss(lrt, lft)
    assert abs(loss - losst.item()) < 1e-12
    losst.backward()
    g_real, g_fake = train.calculate_discriminator_loss_backward(lr, lf, t1, t0)
    G = {}
    d.backward(g_real, cr, G)
    d.backward(g_fake, cf, G)
    for k in d.params:
        assert rel(G[k], P[k].grad.numpy()) < 1e-6, k
    # eval-mode BN uses the running statistics  (srgan_train.py:1228)
    le = d.forward(fake, train=False)
    let = tr.discriminator_forward(P, S, torch.tensor(fake), False)
    assert rel(le, let.detach().numpy()) < 1e-9


@pytest.mark.parametrize("kind", ["gaussian", "uniform"])
def test_generator_loss_and_gradient_vs_torch(kind):
    rs = np.random.RandomState(13)
    y_pred = rs.rand(2, 1, 36, 36)
    y_true = rs.rand(2, 1, 36, 36)
    x = rs.rand(2, 1, 11, 11)
    fake_labels = rs.normal(size=(2, 1))
    x_topo = x[:, :, 1:-1, 1:-1]
    n = 2
    v = train.calculate_generator_loss(y_pred, y_true, fake_labels, np.ones((n, 1)), np.ones((n, 1), np.int32),
                                       np.zeros((n, 1), np.int32), x_topo, ssim_window=kind)
    yp = torch.tensor(y_pred, requires_grad=True)
    vt = tr.g_loss(yp, torch.tensor(y_true), torch.tensor(fake_labels), torch.tensor(x_topo), kind)
    assert abs(v - vt.item()) < 1e-10
    vt.backward()
    g = train.calculate_generator_loss_backward(y_pred, y_true, x_topo, ssim_window=kind)
    assert rel(g, yp.grad.numpy()) < 1e-8


def test_adam_matches_closed_form_first_step():
    """After one step from m=v=0: p -= alpha * g/(|g| + eps*sqrt(1-b2)/...) in Chainer's form."""
    p = np.array([1.0, -2.0, 0.5])
    g = np.array([0.3, -0.1, 2.0])
    m, v = np.zeros(3), np.zeros(3)
    p0 = p.copy()
    ops.adam_update(p, g, m, v, 1, alpha=1e-3, eps=1e-8)
    mh, vh = 0.1 * g, 0.001 * g * g
    alpha_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(p, p0 - alpha_t * mh / (np.sqrt(vh) + 1e-8), rtol=1e-12)


def test_float32_oracle_close_to_float64():
    """The float32 oracle (the thing the HIP path is compared with) tracks float64 to ~1e-5."""
    rs = np.random.RandomState(21)
    g64 = _scaled_generator(2, 5.0, np.float64)
    g32 = _scaled_generator(2, 5.0, np.float32)
    for k in g64.params:
        g32.params[k] = g64.params[k].astype(np.float32)
    ins = _inputs(1, rs)
    y64 = g64.forward(*ins)
    y32 = g32.forward(*(a.astype(np.float32) for a in ins))
    assert rel(y32, y64) < 2e-5
