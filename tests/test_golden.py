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


def test_full_size_fixture_belongs_to_this_oracle():
    """tests/golden/esrgan_full.npz (batch 64 / 16 RRDB / 288 x 288: minutes of oracle time, made by make_golden_full.py)
    is pinned to the current oracle code on the CPU through what is cheap: the generator has no cross-sample coupling, so
    the first two images of the stored forward outputs must be the oracle's outputs for the first two tiles; and every
    array of the three configurations is present with a consistent layout."""
    import make_golden_full as mgf

    gold = dict(np.load(mgf.PATH))
    a = mgf.arrays(64, 4200)
    g, _ = mgf.models_c3()
    y = g.forward(*(a[k][:2] for k in ("X", "W1", "W2", "W3")))
    assert rel(y, gold["c3/g_forward"][:2]) < max(1e-5, 3 * float(gold["c3/g_forward_dev"]))
    a = mgf.arrays(32, 3100)
    y = mgf.models_c2().forward(*(a[k][:2] for k in ("X", "W1", "W2", "W3")))
    assert rel(y, gold["c2/g_forward"][:2]) < 1e-5
    for prefix, shapes in (("c3/gradG/", mgf.omodel.generator_param_shapes(12)), ("c2/gradG/", mgf.omodel.generator_param_shapes(16)),
                           ("c3/gradD/", mgf.omodel.discriminator_param_shapes())):
        stats, offs = gold[prefix + "stats"], gold[prefix + "offsets"]
        sizes = [min(int(np.prod(shapes[k])), mgf.NSAMPLE) for k in sorted(shapes)]
        assert stats.shape == (len(shapes), 2 + mgf.NPROJ) and list(np.diff(offs)) == sizes and offs[-1] == gold[prefix + "samples"].size
        assert np.isfinite(stats).all() and (stats[:, 0] >= 0).all()
    assert gold["c3/gradD/dev"].shape == (len(mgf.omodel.discriminator_param_shapes()), 2)
    assert tuple(gold["c5/shape"]) == (1, 1, 1144, 1144) and gold["c5/grid"].shape == (143, 143)


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


def test_chainer_layout_files_match_the_parameter_tables():
    """tests/golden/chainer_layout_*.npz are written by hand from the layout listing; the oracle's
    parameter tables (built from srgan_train.py's constructor calls) must name exactly the same arrays with the same
    shapes, and `N` must be the 0-d integer Chainer stores."""
    from oracle import model as omodel

    with np.load(os.path.join(HERE, "golden", "chainer_layout_generator.npz")) as f:
        shapes = omodel.generator_param_shapes(2)
        assert set(f.files) == set(shapes)
        for k in f.files:
            assert f[k].shape == tuple(shapes[k]) and f[k].dtype == np.float32, k
    with np.load(os.path.join(HERE, "golden", "chainer_layout_discriminator.npz")) as f:
        shapes = dict(omodel.discriminator_param_shapes())
        shapes.update(omodel.discriminator_persistent_shapes())
        assert set(f.files) == set(shapes)
        for k in f.files:
            assert f[k].shape == tuple(shapes[k]), k
        assert f["batch_norm4/N"].ndim == 0 and f["batch_norm4/N"].dtype.kind == "i" and int(f["batch_norm4/N"]) == 12


@pytest.mark.gpu
def test_chainer_layout_files_load_strictly():
    """deepbedmap.py:402-408: `chainer.serializers.load_npz(file=..., obj=model)` on weight files in Chainer's layout."""
    import deepbedmap_amd as dbm

    g = dbm.GeneratorModel(num_residual_blocks=2, initialize=False)
    d = dbm.DiscriminatorModel(initialize=False)
    gpath = os.path.join(HERE, "golden", "chainer_layout_generator.npz")
    dpath = os.path.join(HERE, "golden", "chainer_layout_discriminator.npz")
    dbm.serializers.load_npz(gpath, g, strict=True)
    dbm.serializers.load_npz(dpath, d, strict=True)
    with np.load(gpath) as f:
        for k in ("input_block/conv_on_W1/W", "residual_network/1/residual_dense_block3/conv_layer5/W", "final_conv_layer2/deform_conv/b"):
            assert np.array_equal(g._tensors[k].array, f[k]), k
    with np.load(dpath) as f:
        for k in ("conv_layer9/W", "batch_norm7/avg_var", "linear_1/W"):
            assert np.array_equal(d._tensors[k].array, f[k]), k
        assert int(d.serialize_dict()["batch_norm4/N"]) == int(f["batch_norm4/N"])
    with pytest.raises(KeyError):  # a 3-RRDB model is not in a 2-RRDB file
        dbm.serializers.load_npz(gpath, dbm.GeneratorModel(num_residual_blocks=3, initialize=False), strict=True)
    x = np.random.RandomState(0).rand(1, 1, 11, 11).astype(np.float32)
    with dbm.using_config("enable_backprop", False):
        y = g.forward(x, np.zeros((1, 1, 110, 110), np.float32), np.zeros((1, 2, 22, 22), np.float32), x).array
    assert y.shape == (1, 1, 36, 36) and np.isfinite(y).all()
