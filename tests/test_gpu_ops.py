"""-m gpu: every HIP kernel family against the NumPy oracle through the C ABI (op-level entry points).

Tolerance: fp32, `max|a-b| / max|b| <= 1e-4` (BASELINE.json north_star); the MFMA kernels accumulate
in fp32 in a different order than BLAS, so results are compared with that tolerance, not bitwise.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import ops

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def dbm():
    import deepbedmap_amd as d
    from deepbedmap_amd import _lib

    ctx = _lib.default_context()
    return d, _lib, ctx


def dev(d, a):
    return d.to_device(np.ascontiguousarray(a, dtype=np.float32))


CONV_CASES = [
    # N, C, H, W, O, k, stride, pad, ups, lrelu
    (3, 64, 9, 9, 32, 3, 1, 1, 0, 1),      # RDB conv_layer1
    (2, 192, 9, 9, 64, 3, 1, 1, 0, 0),     # RDB conv_layer5
    (5, 128, 9, 9, 64, 3, 1, 1, 0, 1),     # pre_residual (ragged tile: 5*81 = 405 positions)
    (2, 64, 9, 9, 64, 3, 1, 1, 1, 1),      # post_upsample_1 (nearest x2 folded in)
    (1, 64, 18, 18, 64, 3, 1, 1, 1, 1),    # post_upsample_2
    (2, 64, 36, 36, 18, 3, 1, 1, 0, 0),    # offset conv (18 channels padded to 32)
    (3, 64, 36, 36, 64, 4, 2, 1, 0, 0),    # D conv_layer1 k4 s2
    (3, 128, 9, 9, 256, 4, 2, 1, 0, 0),    # D conv_layer5 9 -> 4 (odd input)
    (4, 512, 2, 2, 512, 4, 2, 1, 0, 0),    # D conv_layer9 2 -> 1
    (2, 256, 4, 4, 256, 3, 1, 1, 0, 0),    # D conv_layer6
    (1, 64, 13, 17, 32, 3, 1, 1, 0, 1),    # non-square fully convolutional use (deepbedmap.py:726)
    (2, 576, 6, 6, 64, 1, 1, 0, 0, 1),     # the deformable conv's GEMM (1x1 over 576 columns)
    (2, 1, 11, 11, 32, 3, 1, 0, 0, 0),     # input block conv_on_X (few-channel kernel)
    (2, 1, 110, 110, 32, 30, 10, 0, 0, 0),  # input block conv_on_W1
    (2, 2, 22, 22, 32, 6, 2, 0, 0, 0),     # input block conv_on_W2
    (2, 1, 36, 36, 64, 3, 1, 1, 0, 1),     # D conv_layer0
    # weight-gradient kernel forms: LDS-DMA two-wavefront tasks (9x9 / 4x4 / 2x2 planes), odd tile counts, ragged
    # image groups, many K slices; register-staged wave tasks; the workgroup form with row bands
    (67, 96, 9, 9, 32, 3, 1, 1, 0, 1),     # 3 input tiles (one idle wavefront), 67 images
    (70, 160, 9, 9, 32, 3, 1, 1, 0, 0),    # 5 input tiles
    (64, 32, 9, 9, 64, 3, 1, 1, 0, 0),     # a single input tile, two output tiles
    (33, 256, 4, 4, 512, 3, 1, 1, 0, 0),   # D conv_layer6: several images per band, ragged last band
    (37, 512, 2, 2, 512, 3, 1, 1, 0, 0),   # D conv_layer8
    (9, 64, 18, 18, 128, 3, 1, 1, 0, 0),   # D conv_layer2: whole-image bands do not fit a wavefront's LDS share
    (5, 64, 11, 13, 32, 3, 1, 1, 0, 0),    # odd, non-square plane (contiguous runs not a multiple of 16 bytes)
    # direct form (no LDS staging; rows of >= 16 positions): many K slices through the pair buffers, ragged segments
    # (36 = 4.5 segments, 18 = 2.25, 21 odd), the folded nearest x2 resize, the two tap halves of a 4x4 stride-2 layer
    (24, 64, 36, 36, 64, 3, 1, 1, 0, 0),   # post_upsample_2-sized plain layer, 4 tiles
    (20, 64, 18, 18, 64, 3, 1, 1, 1, 0),   # post_upsample_2: 36 x 36 output from an 18 x 18 input
    (12, 96, 36, 36, 32, 4, 2, 1, 0, 0),   # 4x4 stride 2, 36 -> 18, three input tiles
    (3, 64, 23, 21, 32, 3, 1, 1, 0, 0),    # odd widths: misaligned rows, ragged last segment
    (2, 32, 21, 19, 64, 4, 2, 1, 0, 0),    # odd input of a stride-2 layer: NOT the direct form (2 * OW != W)
    # 1x1 on large planes (LDS-staged double-buffered GEMM): the deformable conv's 576 -> 64 at 36 x 36, ragged last band
    # (1296 = 40.5 bands of 32), ragged channel group (576 = 4.5 x 128), many K slices
    (9, 576, 36, 36, 64, 1, 1, 0, 0, 0),
    (3, 160, 20, 20, 96, 1, 1, 0, 0, 0),   # two output groups (96 = 64 + 32), 400 positions = 12.5 bands
    # 4x4 stride-2 layers on tiny planes (workgroup form, several K slices through the pair buffers), odd channel tiles
    (64, 128, 9, 9, 256, 4, 2, 1, 0, 0),   # D conv_layer5 at the full batch: 1024 positions -> 8 K slices (pair buffers)
    (37, 256, 4, 4, 512, 4, 2, 1, 0, 0),   # D conv_layer7
    # position-major tiles with live taps only (igemm_pm_kernel: planes <= 4 x 4, >= 16 images): whole and ragged 32-image
    # groups, every cross-workgroup K split the launcher picks (1 .. 16), one and two output tiles per wavefront, a 1 x 1 output
    (64, 512, 2, 2, 512, 3, 1, 1, 0, 0),   # D conv_layer8 at the full batch (4 of 9 taps live at every position)
    (64, 512, 2, 2, 512, 4, 2, 1, 0, 0),   # D conv_layer9 (4 of 16)
    (64, 256, 4, 4, 512, 4, 2, 1, 0, 1),   # D conv_layer7 (9 of 16)
    (16, 256, 4, 4, 256, 3, 1, 1, 0, 1),   # half a group
    (40, 64, 3, 2, 64, 3, 1, 1, 0, 0),     # non-square plane, one two-tile output group, no K split possible below 32 channels
    (33, 96, 4, 3, 32, 3, 1, 1, 0, 0),     # one output tile per wavefront; 96 channels: 24 per wavefront (three sets)... the general form
    (48, 128, 1, 1, 96, 3, 1, 1, 0, 0),    # a 1 x 1 plane under a 3x3 kernel: only the centre tap is live
    (64, 96, 2, 2, 160, 4, 2, 1, 0, 0),    # conv_layer9-shaped: 12 of 16 taps never inside the image; 3 x 5 tiles
    # planes ONE pixel wide (the 1x1 view of the input block's branches on 3 x 3 tiles): the weight gradient's contiguous staging
    # decoded positions with ceil(2^32 / width), which does not exist for width 1 (fixed in round 3)
    (1, 96, 1, 1, 32, 1, 1, 0, 0, 0),
    (3, 96, 2, 1, 32, 1, 1, 0, 0, 0),
    (2, 64, 5, 1, 64, 3, 1, 1, 0, 1),
    # LDS-tiled form (conv_tile.hip, round 5): 3x3 / pad 1 on 36 x 36 and 18 x 18 OUTPUT planes -- four bands of nine rows (36), the
    # whole image per workgroup (18, >= 256 workgroups) or two bands (18, fewer); ragged output-channel tiles, 32..160 input channels
    (64, 64, 18, 18, 128, 3, 1, 1, 0, 1),  # D conv_layer2 at the full batch: 64 x 4 = 256 whole-image workgroups
    (3, 96, 18, 18, 64, 3, 1, 1, 0, 0),    # two bands per image
    (2, 32, 36, 36, 64, 3, 1, 1, 0, 1),    # the data gradient of an offset convolution's shape (32 padded gradient channels)
    (2, 160, 36, 36, 96, 3, 1, 1, 0, 0),   # twenty chunks of eight channels, three output tiles
    (3, 128, 9, 9, 64, 3, 1, 1, 1, 1),     # 9 x 9 folded x2 -> 18 x 18 output (post_upsample_conv_layer_1)
    (5, 128, 18, 18, 128, 4, 2, 1, 0, 1),  # 4x4 stride 2, 18 -> 9 (discriminator conv_layer3): whole-image workgroups, three tiles
    (7, 64, 36, 36, 96, 4, 2, 1, 0, 0),    # 4x4 stride 2, 36 -> 18 (conv_layer1): two bands, four-channel chunks, three output tiles
    (64, 128, 9, 9, 128, 3, 1, 1, 0, 1),   # 3x3 on 9 x 9 planes at the full batch (conv_layer4): one image per workgroup
]


def _random_conv_cases(seed, n):
    """Randomised geometry on top of the hand-picked cases: every layer kind of the models (3x3 pad 1 with and without the folded
    nearest x2 resize, 4x4 stride 2 pad 1, 1x1) on planes from one pixel up, one to six images, 32..256 input channels."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        kind = int(rs.randint(0, 4))
        N, C = int(rs.randint(1, 7)), 32 * int(rs.randint(1, 9))
        O = int(rs.choice([32, 64, 96, 128]))
        H, W = int(rs.randint(1, 41)), int(rs.randint(1, 41))
        if kind == 0:
            out.append((N, C, H, W, O, 3, 1, 1, 0, int(rs.randint(0, 2))))
        elif kind == 1:
            out.append((N, C, max(1, H // 2), max(1, W // 2), O, 3, 1, 1, 1, int(rs.randint(0, 2))))
        elif kind == 2:
            out.append((N, C, max(2, H), max(2, W), O, 4, 2, 1, 0, 0))
        else:
            out.append((N, C, H, W, O, 1, 1, 0, 0, int(rs.randint(0, 2))))
    return out


CONV_CASES += _random_conv_cases(404, 14)


def _random_tail_cases(seed, n):
    """The deep-discriminator forms at random geometry (igemm_pm_kernel: position-major tiles, live taps only; wgrad_s2tiny_kernel
    for 3x3 and 4x4 stride-2 layers): 16..80 images (whole, ragged and half-empty 32-image groups), planes of 1 x 1 .. 4 x 4 pixels,
    square or not, 64..512 input channels (every K split the launchers pick), 32..256 output channels (one / two tiles per
    wavefront, ragged last tile)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        N = int(rs.randint(16, 81))
        C = int(rs.choice([64, 128, 192, 256, 512]))
        O = int(rs.choice([32, 64, 96, 128, 256]))
        H, W = int(rs.randint(1, 5)), int(rs.randint(1, 5))
        if rs.randint(0, 2):
            out.append((N, C, H, W, O, 3, 1, 1, 0, int(rs.randint(0, 2))))
        else:
            out.append((N, C, max(2, H), max(2, W), O, 4, 2, 1, 0, int(rs.randint(0, 2))))
    return out


CONV_CASES += _random_tail_cases(77, 12)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward(dbm, case):
    d, _lib, ctx = dbm
    N, Cc, H, W, O, k, s, p, ups, act = case
    rs = np.random.RandomState(hash(case) % 2**31)
    x = rs.normal(size=(N, Cc, H, W)).astype(np.float32)
    w = (rs.normal(size=(O, Cc, k, k)) / np.sqrt(Cc * k * k)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    xin = ops.upsample_nearest2(x) if ups else x
    ref = ops.conv2d(xin, w, b, s, p)
    if act:
        ref = ops.leaky_relu(ref)
    y = d.DeviceArray(ref.shape)
    dx, dw, db = dev(d, x), dev(d, w), dev(d, b)  # keep the device buffers alive across the call
    _lib.check(_lib.lib().dbm_op_conv2d(ctx.handle, dx.ptr, dw.ptr, db.ptr, y.ptr, N, Cc, H, W, O, k, s, p, ups, act),
               ctx.handle)
    assert rel(y.get(), ref) < TOL


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_backward(dbm, case):
    d, _lib, ctx = dbm
    N, Cc, H, W, O, k, s, p, ups, _ = case
    rs = np.random.RandomState(hash(case) % 2**31 + 1)
    x = rs.normal(size=(N, Cc, H, W)).astype(np.float32)
    w = (rs.normal(size=(O, Cc, k, k)) / np.sqrt(Cc * k * k)).astype(np.float32)
    xin = ops.upsample_nearest2(x) if ups else x
    OH = (xin.shape[2] + 2 * p - k) // s + 1
    OW = (xin.shape[3] + 2 * p - k) // s + 1
    # the data gradient needs O % 32 == 0 (the models pad the 18 offset channels to 32)
    O_pad = O if O % 32 == 0 else 32
    gy = np.zeros((N, O_pad, OH, OW), np.float32)
    gy[:, :O] = rs.normal(size=(N, O, OH, OW))
    w_pad = np.zeros((O_pad, Cc, k, k), np.float32)
    w_pad[:O] = w
    gx_ref, gw_ref, gb_ref = ops.conv2d_backward(xin, w_pad, gy, s, p)
    want_gx = Cc % 32 == 0
    gx = d.DeviceArray(xin.shape) if want_gx else None
    gw = dev(d, np.zeros_like(w_pad))
    gb = dev(d, np.zeros(O_pad, np.float32))
    dx, dw, dgy = dev(d, x), dev(d, w_pad), dev(d, gy)
    _lib.check(_lib.lib().dbm_op_conv2d_backward(ctx.handle, dx.ptr, dw.ptr, dgy.ptr, gx.ptr if want_gx else None,
                                                 gw.ptr, gb.ptr, N, Cc, H, W, O_pad, k, s, p, ups), ctx.handle)
    ctx.synchronize()
    assert rel(gw.get(), gw_ref) < TOL
    assert rel(gb.get(), gb_ref) < TOL
    if want_gx:
        assert rel(gx.get(), gx_ref) < TOL


@pytest.mark.parametrize("O,scale,shape", [(64, 0.3, (2, 12, 10)), (1, 0.3, (2, 12, 10)), (64, 3.0, (2, 12, 10)), (1, 3.0, (2, 12, 10)),
                                           # the model's own plane: 3 x 1296 positions = 60.75 tiles of the fused kernels (ragged
                                           # last workgroup, tiles straddling images), CSR lists of a full 36 x 36 plane
                                           (64, 1.0, (3, 36, 36)), (1, 1.0, (3, 36, 36))])
def test_deform_conv_forward_backward(dbm, O, scale, shape):
    """final_conv_layer1 (64->64) and final_conv_layer2 (64->1); scale 3.0 drives samples out of the image
    so that the border clipping and the coordinate-gradient masks are exercised."""
    d, _lib, ctx = dbm
    (N, H, W), Cc = shape, 64
    rs = np.random.RandomState(int(scale * 10) + O)
    x = rs.normal(size=(N, Cc, H, W)).astype(np.float32)
    off = rs.normal(scale=scale, size=(N, 18, H, W)).astype(np.float32)
    w = (rs.normal(size=(O, Cc, 3, 3)) / np.sqrt(Cc * 9)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    ref = ops.deform_conv2d(x, off, w, b)
    y = d.DeviceArray(ref.shape)
    l = _lib.lib()
    dx, doff, dw, db = dev(d, x), dev(d, off), dev(d, w), dev(d, b)
    _lib.check(l.dbm_op_deform_conv2d(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y.ptr, N, Cc, H, W, O), ctx.handle)
    assert rel(y.get(), ref) < TOL
    # the generator's other forms of the same forward pass: 64 -> 1 with the multiplication before the sampler (fp32: the same
    # tolerance), 64 -> 64 in split-bf16 arithmetic (sixteen significand bits per operand: 3e-5 of the output's largest value)
    y2 = d.DeviceArray(ref.shape)
    _lib.check(l.dbm_op_deform_conv2d_form(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y2.ptr, N, H, W, O, 1 if O == 1 else 2, 0), ctx.handle)
    assert rel(y2.get(), ref) < (TOL if O == 1 else 3e-5)
    assert not np.array_equal(y2.get(), y.get())  # (a different kernel did run)
    gy = rs.normal(size=ref.shape).astype(np.float32)
    gx_ref, goff_ref, gw_ref, gb_ref = ops.deform_conv2d_backward(x, off, w, gy)
    gx, goff = d.DeviceArray(x.shape), d.DeviceArray(off.shape)
    gw, gb = dev(d, np.zeros_like(w)), dev(d, np.zeros_like(b))
    dgy = dev(d, gy)
    _lib.check(l.dbm_op_deform_conv2d_backward(ctx.handle, dx.ptr, doff.ptr, dw.ptr, dgy.ptr, gx.ptr, goff.ptr, gw.ptr,
                                               gb.ptr, N, Cc, H, W, O), ctx.handle)
    ctx.synchronize()
    assert rel(gx.get(), gx_ref) < TOL
    assert rel(goff.get(), goff_ref) < 5e-4  # bilinear-gradient sums of +/- terms: a little looser
    assert rel(gw.get(), gw_ref) < TOL
    assert rel(gb.get(), gb_ref) < TOL


@pytest.mark.parametrize("shape,scale,lrelu", [((1, 40, 56), 0.5, 1),     # offsets well inside the LDS window
                                               ((2, 37, 21), 1.0, 0),     # ragged tiles, two images, a few samples past the window
                                               ((1, 64, 64), 4.0, 1),     # most taps leave the window: the global fallback
                                               ((1, 5, 3), 0.3, 0),       # a plane smaller than one tile
                                               ((1, 130, 140), 0.8, 1)])  # 81 tiles: the XCD-contiguous tile order
def test_deform_conv_x3_window_kernel_is_bitwise_the_gathering_kernel(dbm, shape, scale, lrelu):
    """The 64 -> 64 deformable layer of the bf16 sweep (srgan_train.py:572) in split-bf16 arithmetic: the kernel whose sampler reads an
    LDS window of the input (form 3; what the sweep's full-resolution planes take) against the kernel that gathers every corner from
    memory (form 4) -- same blend, same split, same summation order: the same bits, whatever the offsets -- and against the oracle."""
    d, _lib, ctx = dbm
    N, H, W = shape
    rs = np.random.RandomState(int(scale * 10) + H + W)
    x = rs.normal(size=(N, 64, H, W)).astype(np.float32)
    off = rs.normal(scale=scale, size=(N, 18, H, W)).astype(np.float32)
    w = (rs.normal(size=(64, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32)
    b = rs.normal(size=(64,)).astype(np.float32)
    ref = ops.deform_conv2d(x, off, w, b)
    if lrelu:
        ref = np.where(ref >= 0, ref, np.float32(0.2) * ref)
    l = _lib.lib()
    dx, doff, dw, db = dev(d, x), dev(d, off), dev(d, w), dev(d, b)
    y3, y4 = d.DeviceArray(ref.shape), d.DeviceArray(ref.shape)
    _lib.check(l.dbm_op_deform_conv2d_form(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y3.ptr, N, H, W, 64, 3, lrelu), ctx.handle)
    _lib.check(l.dbm_op_deform_conv2d_form(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y4.ptr, N, H, W, 64, 4, lrelu), ctx.handle)
    assert rel(y4.get(), ref) < 1e-4
    assert np.array_equal(y3.get(), y4.get())


def test_deform1_premultiplication_on_the_matrix_pipes(dbm):
    """The 64 -> 1 deformable layer (srgan_train.py:574) on a plane of the area sweep's size class (>= 2^18 positions): its
    premultiplication -- the 1x1 convolution 64 -> 9 tap planes -- then runs on v_mfma_f32_16x16x4f32 (deform1_premul_mfma_kernel; the
    training tile's planes keep the vector-ALU kernel and its summation order).  Against the oracle, fp32 tolerance."""
    d, _lib, ctx = dbm
    N, H, W = 1, 512, 520
    rs = np.random.RandomState(77)
    x = rs.normal(size=(N, 64, H, W)).astype(np.float32)
    off = rs.normal(scale=0.7, size=(N, 18, H, W)).astype(np.float32)
    w = (rs.normal(size=(1, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32)
    b = rs.normal(size=(1,)).astype(np.float32)
    ref = ops.deform_conv2d(x, off, w, b)
    dx, doff, dw, db = dev(d, x), dev(d, off), dev(d, w), dev(d, b)
    y = d.DeviceArray(ref.shape)
    _lib.check(_lib.lib().dbm_op_deform_conv2d_form(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y.ptr, N, H, W, 1, 1, 0), ctx.handle)
    assert rel(y.get(), ref) < TOL


def test_loss_known_answers(dbm):
    """The reference's doctest values through the HIP loss kernels (fp32)."""
    d, _, _ = dbm
    v = d.calculate_discriminator_loss(
        real_labels_pred=d.Variable(np.array([[1.1], [-0.5]])), fake_labels_pred=d.Variable(np.array([[-0.3], [1.0]])),
        real_minus_fake_target=np.array([[1], [1]]), fake_minus_real_target=np.array([[0], [0]]))
    assert abs(float(v) - 1.56670504) < 2e-6  # srgan_train.py:985-991
    g = d.calculate_generator_loss(
        y_pred=d.Variable(np.ones(shape=(2, 1, 12, 12))), y_true=np.full(shape=(2, 1, 12, 12), fill_value=10.0),
        fake_labels=np.array([[-1.2], [0.5]]), real_labels=np.array([[0.5], [-0.8]]),
        fake_minus_real_target=np.array([[1], [1]]).astype(np.int32),
        real_minus_fake_target=np.array([[0], [0]]).astype(np.int32), x_topo=np.full(shape=(2, 1, 3, 3), fill_value=9.0))
    assert abs(float(g) - 4.35108415) < 1e-5  # srgan_train.py:859-868
    assert abs(d.psnr(np.ones((2, 1, 3, 3)), np.full((2, 1, 3, 3), 2)) - 192.65919722494797) < 1e-3  # :916-920
    s = d.ssim_loss_func(d.Variable(np.ones((2, 1, 9, 9))), np.full((2, 1, 9, 9), 2.0))
    assert abs(float(s) - 0.800004) < 1e-6  # :944-948
    with pytest.raises(ValueError):  # :950-951
        d.ssim_loss_func(d.Variable(np.ones((2, 1, 9, 9))), np.ones((2, 1, 10, 10)))


@pytest.mark.parametrize("window", ["gaussian", "uniform"])
def test_generator_loss_and_gradient(dbm, window):
    d, _lib, ctx = dbm
    from oracle import train as otrain

    rs = np.random.RandomState(3)
    n = 5
    y = rs.rand(n, 1, 36, 36).astype(np.float32)
    t = rs.rand(n, 1, 36, 36).astype(np.float32)
    X = rs.rand(n, 1, 11, 11).astype(np.float32)
    fl = rs.normal(size=(n, 1)).astype(np.float32)
    # float64 oracle: E[x^2]-mu^2 in fp32 is the noisy side (the HIP kernel works on mean-shifted tiles instead)
    y64, t64, X64 = y.astype(np.float64), t.astype(np.float64), X.astype(np.float64)
    ref = otrain.calculate_generator_loss(y64, t64, fl.astype(np.float64), np.ones((n, 1)), np.ones((n, 1), np.int32),
                                          np.zeros((n, 1), np.int32), X64[:, :, 1:-1, 1:-1], ssim_window=window)
    gref = otrain.calculate_generator_loss_backward(y64, t64, X64[:, :, 1:-1, 1:-1], ssim_window=window)
    out = np.empty(3, np.float32)
    gy = np.empty_like(y)
    wts = (C.c_float * 4)(1e-2, 2e-2, 2e-3, 5.25)
    _lib.check(_lib.lib().dbm_generator_loss(ctx.handle, y.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                                             X.ctypes.data_as(C.c_void_p), None, fl.ctypes.data_as(C.c_void_p), n, 36,
                                             36, wts, 0, 1, {"gaussian": 0, "uniform": 1}[window],
                                             out.ctypes.data_as(C.c_void_p), gy.ctypes.data_as(C.c_void_p), 0),
               ctx.handle)
    assert abs(out[0] - ref) / abs(ref) < 1e-5
    assert abs(out[1] - ops.psnr(y, t)) < 1e-3
    assert abs(out[2] - ops.ssim(y, t, kind=window)) < 1e-5
    assert rel(gy, gref) < TOL


@pytest.mark.parametrize("O", [1, 3])
def test_deform_conv_premultiplied_form_at_dem_range(dbm, O):
    """The last layer of the generator at the reference's data range (un-normalised metres, offsets of a fraction of a pixel
    up to two pixels), O = out_channels 1 and 3: premultiplied tap planes + scalar gathers against the float64 oracle --
    1e-5 of the output's range (the re-association costs nothing measurable)."""
    d, _lib, ctx = dbm
    N, H, W = 2, 40, 37
    rs = np.random.RandomState(70 + O)
    x = (rs.normal(size=(N, 64, H, W)) * 700.0 + 500.0).astype(np.float32)
    off = rs.normal(scale=0.7, size=(N, 18, H, W)).astype(np.float32)
    w = (rs.normal(size=(O, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    ref = ops.deform_conv2d(x.astype(np.float64), off.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    y = d.DeviceArray(ref.shape)
    dx, doff, dw, db = dev(d, x), dev(d, off), dev(d, w), dev(d, b)
    _lib.check(_lib.lib().dbm_op_deform_conv2d_form(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y.ptr, N, H, W, O, 1, 0), ctx.handle)
    assert rel(y.get(), ref) < 1e-5


def _random_deform_cases(seed, n):
    rs = np.random.RandomState(seed)
    return [(int(rs.randint(1, 4)), int(rs.randint(1, 30)), int(rs.randint(1, 30)), int(rs.choice([1, 2, 5, 16, 64])), float(rs.choice([0.2, 1.5, 6.0])))
            for _ in range(10)]


@pytest.mark.parametrize("case", _random_deform_cases(303, 10))
def test_deform_conv_forms_random_shapes(dbm, case):
    """Randomised geometry for the generator's forward forms of the deformable layers (premultiplied few-channel form for
    O <= 16, split-bf16 form for O = 64): planes from a single pixel on, offsets from a fraction of a pixel to far outside the
    image, against the oracle AND against the sampler-then-GEMM kernel of the same library."""
    d, _lib, ctx = dbm
    N, H, W, O, scale = case
    rs = np.random.RandomState(int(scale * 10) + N + H + W + O)
    x = rs.normal(size=(N, 64, H, W)).astype(np.float32)
    off = rs.normal(scale=scale, size=(N, 18, H, W)).astype(np.float32)
    w = (rs.normal(size=(O, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32)
    b = rs.normal(size=(O,)).astype(np.float32)
    ref = ops.deform_conv2d(x, off, w, b)
    l = _lib.lib()
    dx, doff, dw, db = dev(d, x), dev(d, off), dev(d, w), dev(d, b)
    y0, y1 = d.DeviceArray(ref.shape), d.DeviceArray(ref.shape)
    _lib.check(l.dbm_op_deform_conv2d(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y0.ptr, N, 64, H, W, O), ctx.handle)
    _lib.check(l.dbm_op_deform_conv2d_form(ctx.handle, dx.ptr, doff.ptr, dw.ptr, db.ptr, y1.ptr, N, H, W, O, 1 if O <= 16 else 2, 0), ctx.handle)
    assert rel(y0.get(), ref) < TOL
    assert rel(y1.get(), ref) < (TOL if O <= 16 else 1e-4)
