"""The oracle against every known answer the reference's own tests hold for the hot path
(SURVEY.md section 8c).  Values are the literal doctest outputs in /root/reference/srgan_train.py."""
import numpy as np
import pytest

from oracle import model, ops, train


def test_discriminator_loss_kat():  # srgan_train.py:985-991
    v = train.calculate_discriminator_loss(
        real_labels_pred=np.array([[1.1], [-0.5]]),
        fake_labels_pred=np.array([[-0.3], [1.0]]),
        real_minus_fake_target=np.array([[1], [1]]),
        fake_minus_real_target=np.array([[0], [0]]),
    )
    assert f"{v:.8f}" == "1.56670504"


@pytest.mark.parametrize("window", ["gaussian", "uniform"])
def test_generator_loss_kat(window):  # srgan_train.py:859-868
    v = train.calculate_generator_loss(
        y_pred=np.ones(shape=(2, 1, 12, 12)),
        y_true=np.full(shape=(2, 1, 12, 12), fill_value=10.0),
        fake_labels=np.array([[-1.2], [0.5]]),
        real_labels=np.array([[0.5], [-0.8]]),
        fake_minus_real_target=np.array([[1], [1]]).astype(np.int32),
        real_minus_fake_target=np.array([[0], [0]]).astype(np.int32),
        x_topo=np.full(shape=(2, 1, 3, 3), fill_value=9.0),
        ssim_window=window,
    )
    assert f"{v:.8f}" == "4.35108415"


def test_psnr_kat():  # srgan_train.py:916-920
    assert ops.psnr(np.ones((2, 1, 3, 3)), np.full((2, 1, 3, 3), 2)) == 192.65919722494797


@pytest.mark.parametrize("window", ["gaussian", "uniform"])
def test_ssim_kat(window):  # srgan_train.py:944-948
    v = ops.ssim(np.ones((2, 1, 9, 9)), np.full((2, 1, 9, 9), 2.0), kind=window)
    assert f"{v:.6f}" == "0.800004"


def test_ssim_shape_mismatch_raises():  # srgan_train.py:950-951
    with pytest.raises(ValueError):
        ops.ssim(np.ones((2, 1, 9, 9)), np.ones((2, 1, 10, 10)))


def test_generator_shape_and_param_count():  # srgan_train.py:437-447
    g = model.GeneratorModel()
    rs = np.random.RandomState(0)
    y = g.forward(
        x=rs.rand(1, 1, 11, 11).astype("float32"),
        w1=rs.rand(1, 1, 110, 110).astype("float32"),
        w2=rs.rand(1, 2, 22, 22).astype("float32"),
        w3=rs.rand(1, 1, 11, 11).astype("float32"),
    )
    assert y.shape == (1, 1, 36, 36)
    assert g.count_params() == 8907749
    assert model.GeneratorModel(num_residual_blocks=16).count_params() == 11785445


def test_discriminator_shape_and_param_count():  # srgan_train.py:601-608
    d = model.DiscriminatorModel()
    y = d.forward(np.random.RandomState(0).rand(2, 1, 36, 36).astype("float32"))
    assert y.shape == (2, 1)
    assert d.count_params() == 10370761


def test_output_is_4x_of_input_minus_2():  # features/steps/test_deepbedmap.py:35-39
    g = model.GeneratorModel(num_residual_blocks=1)
    rs = np.random.RandomState(1)
    h, w = 14, 17
    y = g.forward(rs.rand(1, 1, h, w).astype("f"), rs.rand(1, 1, 10 * h, 10 * w).astype("f"),
                  rs.rand(1, 2, 2 * h, 2 * w).astype("f"), rs.rand(1, 1, h, w).astype("f"))
    assert y.shape[2] / (h - 2) == 4 and y.shape[3] / (w - 2) == 4


def _fixture_arrays(n=2):  # srgan_train.py:1100-1106
    return {
        "X": np.random.RandomState(seed=42).rand(n, 1, 11, 11).astype(np.float32),
        "W1": np.random.RandomState(seed=42).rand(n, 1, 110, 110).astype(np.float32),
        "W2": np.random.RandomState(seed=42).rand(n, 2, 22, 22).astype(np.float32),
        "W3": np.random.RandomState(seed=42).rand(n, 1, 11, 11).astype(np.float32),
        "Y": np.random.RandomState(seed=42).rand(n, 1, 36, 36).astype(np.float32),
    }


def test_d_step_changes_weights():  # srgan_train.py:1107-1122
    arrays = _fixture_arrays()
    d = model.DiscriminatorModel()
    g = model.GeneratorModel(num_residual_blocks=2)
    opt = train.Adam(d.params, alpha=0.001, eps=1e-7)
    names = model.chainer_param_order(list(d.params))
    w0 = d.params[names[-3]].ravel()[0].copy()
    loss, accu = train.train_eval_discriminator(arrays, g, d, opt)
    w1 = d.params[names[-3]].ravel()[0]
    assert names[-3] == "linear_1/b"
    assert w0 != w1 and np.isfinite(loss) and 0.0 <= accu <= 1.0


def test_g_step_changes_weights():  # srgan_train.py:1197-1212
    arrays = _fixture_arrays()
    g = model.GeneratorModel(num_residual_blocks=2)
    d = model.DiscriminatorModel()
    opt = train.Adam(g.params, alpha=0.001, eps=1e-7)
    names = model.chainer_param_order(list(g.params))
    assert names[8] == "input_block/conv_on_W1/W"
    w0 = g.params[names[8]][0, 0, 0, 0].copy()
    out = train.train_eval_generator(arrays, g, d, opt)
    assert w0 != g.params[names[8]][0, 0, 0, 0]
    assert all(np.isfinite(v) for v in out)


def test_train_requires_optimizer():  # srgan_train.py:1126-1127, 1217-1218
    arrays = _fixture_arrays()
    g = model.GeneratorModel(num_residual_blocks=1)
    d = model.DiscriminatorModel()
    with pytest.raises(AssertionError):
        train.train_eval_discriminator(arrays, g, d, None, train=True)
    with pytest.raises(AssertionError):
        train.train_eval_generator(arrays, g, d, None, train=True)
