"""-m gpu: the parts of the reference's signatures that round 2 still refused ("API narrowings") and the
loss / metric kernels at the REFERENCE'S DATA RANGE (raw metres), against the oracle.

* per-sample int32 targets of calculate_discriminator_loss / calculate_generator_loss (srgan_train.py:960-1004 hands them to
  F.sigmoid_cross_entropy: any array of 0 / 1 / -1);
* ssim_loss_func(window_size, stride) (srgan_train.py:932-956);
* GeneratorModel(out_channels > 1) (srgan_train.py:450-457), forward only;
* load_trained_model: num_residual_blocks from the .npz keys (the reference does not serialise it, srgan_train.py:459-460).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import model as omodel
from oracle import ops
from oracle import train as otrain

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_full as mgf  # noqa: E402

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def dbm():
    import deepbedmap_amd as d

    return d


@pytest.fixture(autouse=True)
def _reset_config(dbm):
    dbm.global_config.train = True
    dbm.global_config.enable_backprop = True
    dbm.global_config.ssim_window = "gaussian"
    dbm.global_config.dtype = "float32"
    yield


def test_per_sample_targets_in_both_losses(dbm):
    rs = np.random.RandomState(11)
    n = 37
    real = rs.normal(size=(n, 1)).astype(np.float32) * 2
    fake = rs.normal(size=(n, 1)).astype(np.float32) * 2
    for t_rf, t_fr in ((rs.randint(0, 2, (n, 1)), rs.randint(0, 2, (n, 1))),          # mixed 0 / 1
                       (rs.randint(-1, 2, (n, 1)), rs.randint(-1, 2, (n, 1))),        # with ignored samples (-1)
                       (np.full((n, 1), -1), np.ones((n, 1), np.int64))):             # one call ignores everything: count -> max(0, 1)
        t_rf, t_fr = t_rf.astype(np.int32), t_fr.astype(np.int32)
        v = dbm.calculate_discriminator_loss(dbm.Variable(real), dbm.Variable(fake), t_rf, t_fr)
        ref = otrain.calculate_discriminator_loss(real.astype(np.float64), fake.astype(np.float64), t_rf, t_fr)
        assert abs(float(v) - ref) < 2e-6 * max(1.0, abs(ref)), (float(v), ref)
        # the gradient w.r.t. the logits (what d_loss.backward() feeds the discriminator) through the C ABI
        from deepbedmap_amd import _lib

        ctx = _lib.default_context()
        out, gr, gf = np.empty(2, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(_lib.lib().dbm_discriminator_loss_t(ctx.handle, p(real), p(fake), n, p(t_rf), p(t_fr), p(out), p(gr), p(gf), 0),
                   ctx.handle)
        gr_ref, gf_ref = otrain.calculate_discriminator_loss_backward(real.astype(np.float64), fake.astype(np.float64), t_rf, t_fr)
        assert np.abs(gr - gr_ref.ravel()).max() < 1e-7 and np.abs(gf - gf_ref.ravel()).max() < 1e-7
        # generator loss: the same targets in its adversarial term
        y = rs.rand(n, 1, 12, 12).astype(np.float32)
        t = rs.rand(n, 1, 12, 12).astype(np.float32)
        xt = rs.rand(n, 1, 3, 3).astype(np.float32)
        g = dbm.calculate_generator_loss(y_pred=dbm.Variable(y), y_true=t, fake_labels=fake, real_labels=real,
                                         fake_minus_real_target=t_fr, real_minus_fake_target=t_rf, x_topo=xt)
        gref = otrain.calculate_generator_loss(y.astype(np.float64), t.astype(np.float64), fake.astype(np.float64),
                                               real.astype(np.float64), t_fr, t_rf, xt.astype(np.float64))
        assert abs(float(g) - gref) < 1e-5 * abs(gref), (float(g), gref)
    with pytest.raises(ValueError):  # chainer's type check: one target per logit, values in {-1, 0, 1}
        dbm.calculate_discriminator_loss(dbm.Variable(real), dbm.Variable(fake), np.ones((n - 1, 1), np.int32), np.zeros((n, 1), np.int32))
    with pytest.raises(ValueError):
        dbm.calculate_discriminator_loss(dbm.Variable(real), dbm.Variable(fake), np.full((n, 1), 2), np.zeros((n, 1), np.int32))


@pytest.mark.parametrize("window", ["gaussian", "uniform"])
@pytest.mark.parametrize("ws,stride,shape", [(9, 1, (3, 1, 36, 36)), (7, 1, (3, 1, 36, 36)), (11, 2, (2, 1, 36, 36)), (4, 4, (2, 2, 36, 36)),
                                             (3, 1, (2, 1, 9, 13)), (9, 3, (1, 1, 144, 100)), (36, 1, (2, 1, 36, 36))])
def test_ssim_window_size_and_stride(dbm, window, ws, stride, shape):
    rs = np.random.RandomState(ws * 31 + stride)
    y = rs.rand(*shape).astype(np.float32)
    t = (0.6 * y + 0.4 * rs.rand(*shape)).astype(np.float32)
    with dbm.using_config("ssim_window", window):
        got = float(dbm.ssim_loss_func(dbm.Variable(y), t, window_size=ws, stride=stride))
    ref = float(ops.ssim(y.astype(np.float64), t.astype(np.float64), ws, stride, window))
    assert abs(got - ref) < 2e-5, (got, ref)
    with pytest.raises(ValueError):
        dbm.ssim_loss_func(dbm.Variable(y), t, window_size=max(shape[2:]) + 1)


@pytest.mark.parametrize("oc", [3, 16])   # (16: the premultiplied 64 -> out_channels layer needs 72 KB of dynamic LDS -- ADVICE round 3)
def test_generator_out_channels_forward_only(dbm, oc):
    og = omodel.GeneratorModel(num_residual_blocks=1, out_channels=oc, seed=21)
    for k in og.params:
        og.params[k] = (og.params[k] * np.float32(5.0) if k.endswith("/W") else
                        np.random.RandomState(5).normal(0, 0.1, og.params[k].shape).astype(np.float32))
    g = dbm.GeneratorModel(num_residual_blocks=1, out_channels=oc, initialize=False)
    assert g.count_params() == og.count_params()
    for name, p in g._tensors.items():
        p.array = og.params[name]
    a = mgf.arrays(3, 77)
    ref = og.forward(a["X"], a["W1"], a["W2"], a["W3"])
    with dbm.using_config("enable_backprop", False):
        y = g.forward(a["X"], a["W1"], a["W2"], a["W3"]).array
        yd = g.forward(*[dbm.to_device(a[k]) for k in ("X", "W1", "W2", "W3")]).array.get()
    assert y.shape == (3, oc, 36, 36) and rel(y, ref) < 1e-4 and np.array_equal(y, yd)
    with pytest.raises(ValueError, match="forward-only"):  # the reference's training step fails with it too
        g.forward(a["X"], a["W1"], a["W2"], a["W3"])
    with pytest.raises(ValueError, match="channels"):
        dbm.calculate_generator_loss(y_pred=dbm.Variable(y), y_true=y, fake_labels=np.zeros((3, 1)), real_labels=np.ones((3, 1)),
                                     fake_minus_real_target=np.ones((3, 1), np.int32), real_minus_fake_target=np.zeros((3, 1), np.int32),
                                     x_topo=a["X"][:, :, 1:-1, 1:-1])


def test_load_trained_model_reads_the_block_count_from_the_file(dbm, tmp_path):
    np.random.seed(9)
    g = dbm.GeneratorModel(num_residual_blocks=3, residual_scaling=0.25)
    path = str(tmp_path / "srgan_generator_model_weights.npz")
    dbm.serializers.save_npz(path, g)
    assert dbm.infer_num_residual_blocks(path) == 3
    m = dbm.load_trained_model(path, residual_scaling=0.25)  # deepbedmap.py:402-408 without Comet's experiment parameters
    assert m.num_residual_blocks == 3 and m.count_params() == g.count_params()
    a = mgf.arrays(2, 5)
    with dbm.using_config("enable_backprop", False):
        assert np.array_equal(m.forward(a["X"], a["W1"], a["W2"], a["W3"]).array, g.forward(a["X"], a["W1"], a["W2"], a["W3"]).array)
    with pytest.raises(ValueError, match="3 residual blocks"):
        dbm.load_npz(path, dbm.GeneratorModel(num_residual_blocks=2, initialize=False))
    d = dbm.DiscriminatorModel()
    dpath = str(tmp_path / "d.npz")
    dbm.serializers.save_npz(dpath, d)
    with pytest.raises(ValueError, match="residual_network"):
        dbm.infer_num_residual_blocks(dpath)


# ---- the loss / metric kernels on un-normalised elevations (metres) ----
def _dem_tiles(n, seed):
    """Prediction / target pairs like a half-trained model's: smooth relief of a few hundred metres around a level of up to
    +-2000 m, the prediction off by tens of metres; one tile carries a -5000 m gap-fill block (deepbedmap.py:164-169)."""
    rs = np.random.RandomState(seed)
    base = rs.uniform(-2000, 2000, (n, 1, 1, 1))
    relief = np.cumsum(np.cumsum(rs.normal(size=(n, 1, 36, 36)), axis=2), axis=3) * 4.0
    t = (base + relief).astype(np.float32)
    y = (t + rs.normal(0, 30, t.shape) + rs.uniform(-50, 50, (n, 1, 1, 1))).astype(np.float32)
    X = (base + rs.normal(0, 100, (n, 1, 11, 11))).astype(np.float32)
    X[0, 0, 2:6, 3:8] = -5000.0
    return y, t, X


@pytest.mark.parametrize("window", ["gaussian", "uniform"])
def test_generator_loss_terms_at_the_reference_data_range(dbm, window):
    """calculate_generator_loss, its gradient, PSNR and SSIM on tiles in metres: C1 = 1e-4 / C2 = 9e-4 against means of 10^3
    and variances of 10^4 (the kernel works on mean-shifted tiles; E[x^2] - mu^2 in float32 has no digits left here)."""
    from deepbedmap_amd import _lib

    ctx = _lib.default_context()
    n = 6
    y, t, X = _dem_tiles(n, 3)
    fl = np.random.RandomState(4).normal(size=(n, 1)).astype(np.float32)
    y64, t64, X64 = (v.astype(np.float64) for v in (y, t, X))
    ref = otrain.calculate_generator_loss(y64, t64, fl.astype(np.float64), np.ones((n, 1)), np.ones((n, 1), np.int32),
                                          np.zeros((n, 1), np.int32), X64[:, :, 1:-1, 1:-1], ssim_window=window)
    gref = otrain.calculate_generator_loss_backward(y64, t64, X64[:, :, 1:-1, 1:-1], ssim_window=window)
    out, gy = np.empty(3, np.float32), np.empty_like(y)
    wts = (C.c_float * 4)(1e-2, 2e-2, 2e-3, 5.25)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(_lib.lib().dbm_generator_loss(ctx.handle, p(y), p(t), p(X), None, p(fl), n, 36, 36, wts, 0, 1,
                                             {"gaussian": 0, "uniform": 1}[window], p(out), p(gy), 0), ctx.handle)
    ssim_ref = float(ops.ssim(y64, t64, kind=window))
    assert 0.05 < ssim_ref < 0.999  # a regime where the structural term matters
    assert abs(out[0] - ref) / abs(ref) < 1e-5, (out[0], ref)
    assert abs(out[1] - ops.psnr(y64, t64)) < 1e-3
    assert abs(out[2] - ssim_ref) < 2e-5, (out[2], ssim_ref)
    assert rel(gy, gref) < 1e-4
    # the metric entry points on the same data
    with dbm.using_config("ssim_window", window):
        assert abs(float(dbm.ssim_loss_func(dbm.Variable(y), t)) - ssim_ref) < 2e-5
        assert abs(float(dbm.ssim_loss_func(dbm.Variable(y), t, window_size=7, stride=2)) - float(ops.ssim(y64, t64, 7, 2, window))) < 2e-5
    assert abs(dbm.psnr(y, t) - float(ops.psnr(y64, t64))) < 1e-3


def test_clip_of_resident_grids_on_the_device(dbm):
    """deepbedmap.py:663-665 (`np.clip(a=W_tile, a_min=0.0, a_max=None)` before the sweep) on grids that already live in HBM:
    clip_inputs on DeviceArrays runs dbm_clip_min_f32 in place, with np.clip's treatment of NaN; predict_tiled_resident(clip=True)
    gives the canvas of the host-clipped call."""
    rs = np.random.RandomState(5)
    a = (rs.normal(size=(1, 1, 70, 90)) * 50).astype(np.float32)
    a[0, 0, 3, 4] = np.nan
    a[0, 0, 5, 6] = -0.0
    dev = dbm.to_device(a)
    out = dbm.clip_inputs(dev, dev, dev)[0]
    assert out is dev
    got, ref = dev.get(), np.clip(a, 0.0, None)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.nan_to_num(got), np.nan_to_num(ref))
    # the resident sweep with clip=True == the host-clipped sweep
    np.random.seed(12)
    g = dbm.GeneratorModel(num_residual_blocks=1)
    H, W = 30, 34
    X = rs.rand(1, 1, H, W).astype(np.float32)
    W1 = (rs.rand(1, 1, 10 * H, 10 * W) - 0.3).astype(np.float32)
    W2 = (rs.rand(1, 2, 2 * H, 2 * W) - 0.3).astype(np.float32)
    W3 = (rs.rand(1, 1, H, W) - 0.3).astype(np.float32)
    S = dbm.Shape
    kw = dict(final_shape=S(y=4 * H, x=4 * W), ary_shape=S(y=40, x=40), stride=S(y=40, x=40), xtrapad=S(y=2, x=2))
    ref = dbm.predict_tiled_resident(g, X, *dbm.clip_inputs(W1, W2, W3), **kw)
    got = dbm.predict_tiled_resident(g, X, W1, W2, W3, clip=True, **kw)
    assert np.array_equal(np.nan_to_num(got, nan=-7.0), np.nan_to_num(ref, nan=-7.0))


def _random_ssim_cases(seed, n):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        h, w = int(rs.randint(1, 60)), int(rs.randint(1, 60))
        ws = int(rs.randint(1, min(h, w, 17) + 1))
        out.append((ws, int(rs.randint(1, 6)), (int(rs.randint(1, 4)), int(rs.randint(1, 3)), h, w), str(rs.choice(["gaussian", "uniform"]))))
    return out


@pytest.mark.parametrize("ws,stride,shape,window", _random_ssim_cases(606, 12))
def test_ssim_random_geometry(dbm, ws, stride, shape, window):
    """ssim_loss_func(window_size, stride) on randomised planes (one pixel on), windows from 1 x 1 to the whole plane, strides 1..5,
    one or two channels -- against the float64 oracle."""
    rs = np.random.RandomState(ws * 131 + stride + shape[2])
    y = rs.rand(*shape).astype(np.float32)
    t = (0.5 * y + 0.5 * rs.rand(*shape)).astype(np.float32)
    with dbm.using_config("ssim_window", window):
        got = float(dbm.ssim_loss_func(dbm.Variable(y), t, window_size=ws, stride=stride))
    ref = float(ops.ssim(y.astype(np.float64), t.astype(np.float64), ws, stride, window))
    assert abs(got - ref) < 2e-5, (got, ref)


_FWD_TIMEOUT_SCRIPT = r"""
import sys, warnings, ctypes as C, numpy as np
sys.path.insert(0, sys.argv[1])
import deepbedmap_amd as d
from deepbedmap_amd import _lib
np.random.seed(3)
g = d.GeneratorModel(num_residual_blocks=2)
lib = _lib.lib()
rs = np.random.RandomState(1)
n = 8
xs = [rs.rand(n, c, m * 11, m * 11).astype(np.float32) for c, m in ((1, 1), (1, 10), (2, 2), (1, 1))]
hp = lambda a: a.ctypes.data_as(C.c_void_p)
with d.using_config("enable_backprop", False):
    ref = g.forward(*xs).array.copy()
    # (1) the C ABI: a forward on host pointers that ends while the condition is up returns status 7 (its results are void) ...
    out = np.zeros_like(ref)
    _lib.check(lib.dbm_debug_inject_timeout_async(g.ctx.handle), g.ctx.handle)
    rc = lib.dbm_gen_forward(g._h, n, 11, 11, hp(xs[0]), hp(xs[1]), hp(xs[2]), hp(xs[3]), hp(out), 0)
    assert rc == 7, rc
    assert g.ctx.timeout_info()[0] == 1
    # ... and the re-issued call is valid (layer-by-layer trunk kernels now)
    rc = lib.dbm_gen_forward(g._h, n, 11, 11, hp(xs[0]), hp(xs[1]), hp(xs[2]), hp(xs[3]), hp(out), 0)
    assert rc == 0, rc
    assert np.abs(out - ref).max() <= 2e-5 * np.abs(ref).max()
    # (2) the Python mirror re-issues by itself, with a warning
    _lib.check(lib.dbm_debug_inject_timeout_async(g.ctx.handle), g.ctx.handle)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = g.forward(*xs).array
    assert any("timed out" in str(x.message) for x in w)
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()
# (3) dbm_adam_update behind an event: status 9, NOTHING applied, step counter unchanged; after a fresh backward the update goes through
opt = d.optimizers.Adam(alpha=1e-3, eps=1e-7).setup(g)
before = {k: v.copy() for k, v in g.serialize_dict().items()}
y = g.forward(*xs)
g.cleargrads()
g.backward(np.ones(y.shape, np.float32))
_lib.check(lib.dbm_debug_inject_timeout_async(g.ctx.handle), g.ctx.handle)
g.ctx.synchronize()   # (the raising kernel has RUN: an update that merely overtakes it on the host is the queued-event case, not this one)
try:
    opt.update()
    raise SystemExit("update() did not report the void gradients")
except _lib.DbmError as e:
    assert e.code == 9, e.code
assert opt.t == 0
after = g.serialize_dict()
assert all(np.array_equal(before[k], after[k]) for k in before)
try:  # a blind re-issue would have applied the void pass's gradients: the arena is what it is, but the CONTRACT is "redo the pass"
    y = g.forward(*xs)
    g.cleargrads()
    g.backward(np.ones(y.shape, np.float32))
    opt.update()
except _lib.DbmError as e:
    raise SystemExit(f"valid pass refused: {e}")
assert opt.t == 1
moved = g.serialize_dict()
assert any(not np.array_equal(before[k], moved[k]) for k in before)
# (4) ADVICE round 4: the event is observed -- and the sticky flag cleared -- by a HOST-SYNCHRONISING call between the void backward
# pass (device pointers: it only enqueues) and the update.  The update must still refuse (status 9) until the arena has been cleared.
y = g.forward(*xs)
g.cleargrads()
_lib.check(lib.dbm_debug_inject_timeout_async(g.ctx.handle), g.ctx.handle)
g.backward(d.to_device(np.ones(y.shape, np.float32)))          # enqueued under the raised flag: void
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    with d.using_config("enable_backprop", False):
        _ = g.forward(*xs).array                                # status 7 inside, repeated by the mirror: flag clean again
assert any("timed out" in str(x.message) for x in w)
t0 = opt.t
held = {k: v.copy() for k, v in g.serialize_dict().items()}
try:
    opt.update()
    raise SystemExit("update() applied the gradients of a void pass after a host-synchronising call had cleared the flag")
except _lib.DbmError as e:
    assert e.code == 9, e.code
assert opt.t == t0 and all(np.array_equal(held[k], v) for k, v in g.serialize_dict().items())
y = g.forward(*xs)
g.cleargrads()
g.backward(np.ones(y.shape, np.float32))
opt.update()
assert opt.t == t0 + 1
print("ok")
"""


def test_timeouts_are_observed_by_host_synchronising_forwards_and_by_adam_update(dbm, tmp_path):
    """a persistent kernel that gives up during a forward-only call must not hand garbage to the host with status 0
    (dbm_gen_forward on host pointers now observes the condition after its final synchronisation: status 7, re-issue), and
    dbm_adam_update must not be re-issuable onto the gradients of a void pass (status 9, nothing applied)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fwd_timeout.py"
    script.write_text(_FWD_TIMEOUT_SCRIPT)
    res = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-3000:] + res.stdout[-500:]
