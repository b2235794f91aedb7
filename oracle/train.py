"""NumPy restatement of the losses and the D-step / G-step (srgan_train.py:841-1263).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import numpy as np
from . import ops


def calculate_discriminator_loss(real_labels_pred, fake_labels_pred, real_minus_fake_target, fake_minus_real_target):
    """srgan_train.py:960-1009 (RaGAN)."""
    real_avg = real_labels_pred.mean()
    fake_avg = fake_labels_pred.mean()
    return ops.sigmoid_cross_entropy(real_labels_pred - fake_avg, real_minus_fake_target) + ops.sigmoid_cross_entropy(
        fake_labels_pred - real_avg, fake_minus_real_target
    )


def calculate_discriminator_loss_backward(real, fake, t_rf, t_fr):
    """d loss / d real_logits, d loss / d fake_logits (both means carry gradient)."""
    g1 = ops.sigmoid_cross_entropy_backward(real - fake.mean(), t_rf)
    g2 = ops.sigmoid_cross_entropy_backward(fake - real.mean(), t_fr)
    g_real = g1 - g2.sum() / real.dtype.type(real.size)
    g_fake = g2 - g1.sum() / fake.dtype.type(fake.size)
    return g_real, g_fake


def calculate_generator_loss(
    y_pred, y_true, fake_labels, real_labels, fake_minus_real_target, real_minus_fake_target, x_topo,
    content_loss_weighting=1e-2, adversarial_loss_weighting=2e-2, topographic_loss_weighting=2e-3,
    structural_loss_weighting=5.25e-0, ssim_window="gaussian",
):
    """srgan_train.py:841-902."""
    content = ops.mean_absolute_error(y_pred, y_true)
    adversarial = calculate_discriminator_loss(real_labels, fake_labels, real_minus_fake_target, fake_minus_real_target)
    topo = ops.mean_absolute_error(ops.average_pooling_4x4(y_pred), x_topo)
    structural = 1 - ops.ssim(y_pred, y_true, 9, 1, ssim_window)
    return (
        content_loss_weighting * content
        + adversarial_loss_weighting * adversarial
        + topographic_loss_weighting * topo
        + structural_loss_weighting * structural
    )


def calculate_generator_loss_backward(
    y_pred, y_true, x_topo, content_loss_weighting=1e-2, topographic_loss_weighting=2e-3,
    structural_loss_weighting=5.25e-0, ssim_window="gaussian",
):
    """d g_loss / d y_pred.  The adversarial term has no path to y_pred: the reference
    feeds it `d_model.forward(fake_images).array` (srgan_train.py:1228-1229), a detached array."""
    f = y_pred.dtype.type
    g = f(content_loss_weighting) * ops.mean_absolute_error_backward(y_pred, y_true)
    pooled = ops.average_pooling_4x4(y_pred)
    g = g + f(topographic_loss_weighting) * ops.average_pooling_4x4_backward(
        ops.mean_absolute_error_backward(pooled, x_topo)
    )
    g = g - f(structural_loss_weighting) * ops.ssim_backward(y_pred, y_true, 9, 1, ssim_window)
    return g.astype(y_pred.dtype)


class Adam:
    """chainer.optimizers.Adam(alpha, eps) .setup(link)  srgan_train.py:1043-1048."""

    def __init__(self, params, alpha=1.6e-4, beta1=0.9, beta2=0.999, eps=1e-8):
        self.params = params
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps
        self.t = 0
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    def update(self, grads):
        self.t += 1
        for k, p in self.params.items():
            ops.adam_update(p, grads[k].astype(p.dtype), self.m[k], self.v[k], self.t, self.alpha,
                            self.beta1, self.beta2, self.eps)


def train_eval_discriminator(input_arrays, g_model, d_model, d_optimizer=None, train=True):
    """srgan_train.py:1084-1166."""
    if train:
        assert d_optimizer is not None
    fake_images = g_model.forward(input_arrays["X"], input_arrays["W1"], input_arrays["W2"], input_arrays["W3"])
    real_images = input_arrays["Y"]
    n = len(real_images)
    dt = real_images.dtype
    real_pred, c_real = d_model.forward(real_images, train=train, keep=True)  # :1145  (two separate BN batches)
    fake_pred, c_fake = d_model.forward(fake_images, train=train, keep=True)  # :1146
    t_rf = np.ones((n, 1), dtype=np.int32)
    t_fr = np.zeros((n, 1), dtype=np.int32)
    d_loss = calculate_discriminator_loss(real_pred, fake_pred, t_rf, t_fr)
    pred = np.concatenate([real_pred, fake_pred])
    truth = np.concatenate([np.ones((n, 1), np.int32), np.zeros((n, 1), np.int32)])
    d_accu = ops.binary_accuracy(pred, truth)
    if train:
        g_real, g_fake = calculate_discriminator_loss_backward(real_pred, fake_pred, t_rf, t_fr)
        G = {}
        d_model.backward(g_real.astype(dt), c_real, accumulate_into=G)
        d_model.backward(g_fake.astype(dt), c_fake, accumulate_into=G)
        d_optimizer.update(G)
    return float(d_loss), float(d_accu)


def train_eval_generator(input_arrays, g_model, d_model, g_optimizer=None, train=True, ssim_window="gaussian"):
    """srgan_train.py:1170-1263."""
    if train:
        assert g_optimizer is not None
    fake_images = g_model.forward(input_arrays["X"], input_arrays["W1"], input_arrays["W2"], input_arrays["W3"], keep=train)
    fake_labels = d_model.forward(fake_images, train=False).astype(np.float32)  # :1228-1229 eval-mode BN, detached
    real_images = input_arrays["Y"]
    n = len(real_images)
    real_labels = np.ones((n, 1), dtype=np.float32)  # :1233
    t_fr = np.ones((n, 1), dtype=np.int32)  # :1236
    t_rf = np.zeros((n, 1), dtype=np.int32)  # :1237
    x_topo = input_arrays["X"][:, :, 1:-1, 1:-1]  # :1248
    g_loss = calculate_generator_loss(fake_images, real_images, fake_labels, real_labels, t_fr, t_rf, x_topo,
                                      ssim_window=ssim_window)
    g_psnr = ops.psnr(fake_images, real_images)
    g_ssim = ops.ssim(fake_images, real_images, 9, 1, ssim_window)
    if train:
        gy = calculate_generator_loss_backward(fake_images, real_images, x_topo, ssim_window=ssim_window)
        g_model.backward(gy)
        g_optimizer.update(g_model.grads)
    return float(g_loss), float(g_psnr), float(g_ssim)
