"""Committed golden vectors (tests/golden/esrgan_small.npz, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them; GPU: the HIP path reproduces them."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

GOLD = dict(np.load(os.path.join(HERE, "golden", "esrgan_small.npz")))


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max()


def test_oracle_reproduces_golden_vectors():
    out = mg.compute()
    assert set(out) == set(GOLD)
    for k, v in GOLD.items():
        assert rel(out[k], v) < 1e-5, k  # BLAS summation order may differ between hosts


@pytest.mark.gpu
def test_hip_reproduces_golden_vectors():
    import deepbedmap_amd as dbm

    arrays = mg.fixture_arrays()
    og, od = mg.build_models()

    def hip_models():
        g = dbm.GeneratorModel(num_residual_blocks=mg.N_BLOCKS, initialize=False)
        d = dbm.DiscriminatorModel(initialize=False)
        for name, p in g._tensors.items():
            p.array = og.params[name]
        for name, p in d._tensors.items():
            p.array = od.params[name] if name in od.params else np.asarray(od.persistent[name], np.float32)
        return g, d

    g, d = hip_models()
    dbm.global_config.train = True
    with dbm.using_config("enable_backprop", False):
        y = g.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"]).array
    assert rel(y, GOLD["g_forward"]) < 1e-4
    assert rel(d.forward(arrays["Y"]).array, GOLD["d_logits_train_real"]) < 1e-4
    g, d = hip_models()
    d_opt = dbm.optimizers.Adam(alpha=1e-3, eps=1e-7).setup(d)
    g_opt = dbm.optimizers.Adam(alpha=1e-3, eps=1e-7).setup(g)
    assert np.allclose(dbm.train_eval_discriminator(arrays, g, d, d_opt), GOLD["d_step"], rtol=2e-4, atol=1e-5)
    assert np.allclose(dbm.train_eval_generator(arrays, g, d, g_opt), GOLD["g_step"], rtol=2e-4, atol=1e-5)
    with dbm.using_config("enable_backprop", False):
        y2 = g.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"]).array
    assert rel(y2, GOLD["g_forward_after_step"]) < 5e-3  # after one Adam step (sign-like first update)
