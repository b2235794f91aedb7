"""-m gpu: the channels-last bf16 convolution of the area sweep's trunk (csrc/conv_cl16.hip, BASELINE config 5).

The kernel's contract is exact: operands rounded to nearest-even bf16, products accumulated in fp32.  The oracle's im2col
convolution on PRE-ROUNDED operands (float64 accumulation) is therefore a tight reference -- 2e-5 of the output's largest
magnitude, the order-of-summation noise of an fp32 accumulator over K <= 1728 terms -- not a 3e-2 "bf16 tolerance".
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import model as omodel
from oracle import ops

pytestmark = pytest.mark.gpu


def bf16_round(a):
    """float32 -> nearest-even bfloat16 -> float32 (what v_cvt_pk_bf16_f32 does)."""
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


@pytest.fixture(scope="module")
def dbm():
    import deepbedmap_amd as d
    from deepbedmap_amd import _lib

    return d, _lib, _lib.default_context()


CASES = [
    # N, C, H, W, O, lrelu, residual
    (1, 64, 40, 37, 32, 1, False),     # conv_layer1: ragged tiles in both directions
    (2, 96, 18, 50, 32, 1, False),     # conv_layer2, two images
    (1, 128, 9, 9, 32, 1, False),      # a plane smaller than one tile
    (1, 160, 23, 16, 32, 0, False),    # conv_layer4, exactly one tile column
    (1, 192, 33, 35, 64, 0, True),     # conv_layer5: two output tiles per wavefront, a5 * rs + a0 epilogue
    (1, 64, 286, 286, 32, 1, False),   # the sweep's trunk plane: 234 workgroups of eleven patches
    (1, 192, 286, 286, 64, 0, True),
]


@pytest.mark.parametrize("N,Cc,H,W,O,lrelu,resid", CASES)
def test_cl16_conv_matches_oracle_on_rounded_operands(dbm, N, Cc, H, W, O, lrelu, resid):
    d, _lib, ctx = dbm
    rs = np.random.RandomState(Cc * 7 + H)
    x = rs.normal(size=(N, Cc, H, W)).astype(np.float32) * 3.0
    w = rs.normal(size=(O, Cc, 3, 3)).astype(np.float32) / np.sqrt(9 * Cc)
    b = rs.normal(size=(O,)).astype(np.float32)
    r1 = rs.normal(size=(N, 64, H, W)).astype(np.float32) if resid else None
    s1 = 0.1 if resid else 1.0
    ref = ops.conv2d(bf16_round(x).astype(np.float64), bf16_round(w).astype(np.float64), b.astype(np.float64), 1, 1)
    if resid:
        ref = s1 * ref + r1
    if lrelu:
        ref = np.where(ref >= 0, ref, 0.2 * ref)
    dx, dw, db = d.to_device(x), d.to_device(w), d.to_device(b)
    dr = d.to_device(r1) if resid else None
    y = d.DeviceArray((N, O, H, W))
    _lib.check(_lib.lib().dbm_op_conv2d_cl16(ctx.handle, dx.ptr, dw.ptr, db.ptr, dr.ptr if resid else None, s1, y.ptr, N, Cc, H, W, O,
                                             lrelu), ctx.handle)
    got = y.get()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err


def test_cl16_trunk_equals_the_per_layer_bf16_path(dbm):
    """GeneratorModel.forward in the bf16 sweep mode with the trunk on conv_cl16 (default) against the same mode on the
    per-layer implicit GEMM (DBM_CL16=0): both round the same operands at the same places (the dense block's activations
    to bf16, the residual stream in fp32), so they agree to accumulation-order noise -- on a plane with ragged tiles,
    three RRDBs (every `x` skip of :402), two images."""
    d, _lib, ctx = dbm
    og = omodel.GeneratorModel(num_residual_blocks=3, seed=31)
    r = np.random.RandomState(32)
    for k in sorted(og.params):
        og.params[k] = (og.params[k] * np.float32(3.0) if k.endswith("/W") else r.normal(0, 0.1, og.params[k].shape).astype(np.float32))
    g = d.GeneratorModel(num_residual_blocks=3, initialize=False)
    for name, p in g._tensors.items():
        p.array = og.params[name]
    h, w = 45, 61
    ins = [d.to_device(r.rand(2, c, m * h, m * w).astype(np.float32)) for c, m in ((1, 1), (1, 10), (2, 2), (1, 1))]
    old = os.environ.get("DBM_CL16")
    try:
        with d.using_config("enable_backprop", False):
            y32 = g.forward(*ins).array.get()
            with d.using_config("dtype", "bfloat16"):
                os.environ["DBM_CL16"] = "1"
                y_cl = g.forward(*ins).array.get()
                os.environ["DBM_CL16"] = "0"
                y_ig = g.forward(*ins).array.get()
    finally:
        if old is None:
            os.environ.pop("DBM_CL16", None)
        else:
            os.environ["DBM_CL16"] = old
    scale = np.abs(y32).max()
    assert np.abs(y_cl - y_ig).max() / scale < 1e-4, np.abs(y_cl - y_ig).max() / scale
    e = np.abs(y_cl - y32).max() / scale
    assert 1e-7 < e < 3e-2, e  # really bf16 arithmetic, within the mode's tolerance of the fp32 forward
    assert not np.array_equal(y_cl, y_ig)  # (two different kernels did run)


X3_CASES = [
    # N, H, W (output plane), O, ups, lrelu, planar
    (1, 40, 37, 64, 0, 1, 0),      # ragged tiles
    (2, 36, 52, 64, 1, 1, 0),      # post_upsample: nearest x2 folded into the staging, two images
    (1, 33, 35, 18, 0, 0, 1),      # offset convolution: 18 channel planes
    (1, 572, 572, 64, 1, 1, 0),    # the sweep's first upsampling layer: 286 -> 572
    (1, 520, 530, 18, 0, 0, 1),    # 1 122 tiles of one output-channel tile: the form with two workgroups per CU (tiles of <= 8 patches)
]


@pytest.mark.parametrize("N,H,W,O,ups,lrelu,planar", X3_CASES)
def test_split_bf16_conv_is_fp32_class(dbm, N, H, W, O, ups, lrelu, planar):
    """conv_cl16x3_kernel (three bf16 MFMAs per product) against the float64 oracle on the UN-rounded fp32 operands: 3e-5 of
    the output's largest magnitude -- sixteen significand bits per operand; the plain bf16 kernel on the same data is two
    orders of magnitude further away (asserted: the split is what buys the accuracy, not the test's tolerance)."""
    d, _lib, ctx = dbm
    rs = np.random.RandomState(H * 3 + O)
    Hs, Ws = H >> ups, W >> ups
    x = (rs.normal(size=(N, 64, Hs, Ws)) * 300.0 + 1000.0).astype(np.float32)   # un-normalised, offset data
    w = (rs.normal(size=(O, 64, 3, 3)) / np.sqrt(9 * 64)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    xu = ops.upsample_nearest2(x) if ups else x
    ref = ops.conv2d(xu.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 1, 1)
    if lrelu:
        ref = np.where(ref >= 0, ref, 0.2 * ref)
    dx, dw, db = d.to_device(x), d.to_device(w), d.to_device(b)
    y = d.DeviceArray((N, O, H, W))
    _lib.check(_lib.lib().dbm_op_conv2d_cl16x3(ctx.handle, dx.ptr, dw.ptr, db.ptr, y.ptr, N, H, W, O, ups, lrelu, planar), ctx.handle)
    err = np.abs(y.get() - ref).max() / np.abs(ref).max()
    assert err < 3e-5, err
    if not ups and O in (32, 64):  # the one-MFMA bf16 kernel on the same data
        y1 = d.DeviceArray((N, O, H, W))
        _lib.check(_lib.lib().dbm_op_conv2d_cl16(ctx.handle, dx.ptr, dw.ptr, db.ptr, None, 1.0, y1.ptr, N, 64, H, W, O, lrelu), ctx.handle)
        err1 = np.abs(y1.get() - ref).max() / np.abs(ref).max()
        assert err1 > 30 * err, (err1, err)


def _random_cases(seed, n, make):
    rs = np.random.RandomState(seed)
    return [make(rs) for _ in range(n)]


@pytest.mark.parametrize("case", _random_cases(101, 10, lambda r: (int(r.randint(1, 4)), 32 * int(r.randint(1, 9)), int(r.randint(1, 75)),
                                                                   int(r.randint(1, 75)), int(r.choice([32, 64])), int(r.randint(0, 2)))))
def test_cl16_conv_random_shapes(dbm, case):
    """Randomised geometry for conv_cl16_kernel: planes from a single pixel up to a few tiles, 32..256 input channels, one to three
    images; conv_layer5's residual epilogue whenever the layer has 64 outputs."""
    d, _lib, ctx = dbm
    N, Cc, H, W, O, lrelu = case
    rs = np.random.RandomState(sum(case))
    x = rs.normal(size=(N, Cc, H, W)).astype(np.float32)
    w = (rs.normal(size=(O, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    resid = O == 64
    r1 = rs.normal(size=(N, 64, H, W)).astype(np.float32) if resid else None
    s1 = 0.1 if resid else 1.0
    ref = ops.conv2d(bf16_round(x).astype(np.float64), bf16_round(w).astype(np.float64), b.astype(np.float64), 1, 1)
    if resid:
        ref = s1 * ref + r1
    if lrelu:
        ref = np.where(ref >= 0, ref, 0.2 * ref)
    y = d.DeviceArray((N, O, H, W))
    dx, dw, db = d.to_device(x), d.to_device(w), d.to_device(b)   # (held: a temporary's memory is freed with the temporary)
    dr = d.to_device(r1) if resid else None
    _lib.check(_lib.lib().dbm_op_conv2d_cl16(ctx.handle, dx.ptr, dw.ptr, db.ptr, dr.ptr if resid else None, s1, y.ptr, N, Cc, H, W, O, lrelu),
               ctx.handle)
    assert np.abs(y.get() - ref).max() / max(np.abs(ref).max(), 1e-30) < 2e-5


@pytest.mark.parametrize("O", [32, 64])
def test_cl16_conv_two_workgroups_per_cu_form(dbm, O):
    """conv_cl16_kernel on a launch of several rounds of tiles (four 272 x 272 planes = 1 156 tiles of 16 rows: crops batched per forward):
    the 32-output-channel layers then run as tiles of <= 8 patches with two workgroups per CU; the 64-channel layer (conv_layer5, with
    its residual epilogue) keeps its one-workgroup form -- both on four planes at once."""
    d, _lib, ctx = dbm
    N, Cc, H, W = 4, 32, 272, 272
    rs = np.random.RandomState(O)
    x = rs.normal(size=(N, Cc, H, W)).astype(np.float32)
    w = (rs.normal(size=(O, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    resid = O == 64
    r1 = rs.normal(size=(N, 64, H, W)).astype(np.float32) if resid else None
    s1 = 0.1 if resid else 1.0
    ref = ops.conv2d(bf16_round(x).astype(np.float64), bf16_round(w).astype(np.float64), b.astype(np.float64), 1, 1)
    if resid:
        ref = s1 * ref + r1
    ref = np.where(ref >= 0, ref, 0.2 * ref)
    y = d.DeviceArray((N, O, H, W))
    dx, dw, db = d.to_device(x), d.to_device(w), d.to_device(b)
    dr = d.to_device(r1) if resid else None
    _lib.check(_lib.lib().dbm_op_conv2d_cl16(ctx.handle, dx.ptr, dw.ptr, db.ptr, dr.ptr if resid else None, s1, y.ptr, N, Cc, H, W, O, 1),
               ctx.handle)
    assert np.abs(y.get() - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize("case", _random_cases(202, 10, lambda r: (int(r.randint(1, 3)), int(r.randint(1, 40)), int(r.randint(1, 40)),
                                                                   int(r.randint(1, 65)), int(r.randint(0, 2)), int(r.randint(0, 2)), int(r.randint(0, 2)))))
def test_split_bf16_conv_random_shapes(dbm, case):
    """Randomised geometry for conv_cl16x3_kernel: any output-channel count 1..64, with and without the folded nearest x2
    resize (even output planes then), NHWC-to-NCHW and channel-plane epilogues."""
    d, _lib, ctx = dbm
    N, Hs, Ws, O, ups, lrelu, planar = case
    H, W = Hs << ups, Ws << ups
    rs = np.random.RandomState(sum(case) + 7)
    x = (rs.normal(size=(N, 64, Hs, Ws)) * 100.0 + 50.0).astype(np.float32)
    w = (rs.normal(size=(O, 64, 3, 3)) / np.sqrt(9 * 64)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    xu = ops.upsample_nearest2(x) if ups else x
    ref = ops.conv2d(xu.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 1, 1)
    if lrelu:
        ref = np.where(ref >= 0, ref, 0.2 * ref)
    y = d.DeviceArray((N, O, H, W))
    dx, dw, db = d.to_device(x), d.to_device(w), d.to_device(b)
    _lib.check(_lib.lib().dbm_op_conv2d_cl16x3(ctx.handle, dx.ptr, dw.ptr, db.ptr, y.ptr, N, H, W, O, ups, lrelu, planar), ctx.handle)
    assert np.abs(y.get() - ref).max() / max(np.abs(ref).max(), 1e-30) < 3e-5
