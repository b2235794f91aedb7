"""-m gpu: the HIP path at BASELINE.json's FULL configuration sizes against oracle outputs committed as
tests/golden/esrgan_full.npz (made by tests/golden/make_golden_full.py; the oracle needs minutes at these sizes).

This is the regime the benchmark runs: 64 images = 192 resident workgroups of the persistent trunk kernels with
three bands per image and granule hand-offs, the prefetched G-step forward, merged discriminator weight-gradient
launches, the ordered (cudnn_deterministic) gradient folds.  Tolerances: 1e-4 forward (BASELINE north_star),
5e-4 gradients (sums over 64 x 81 .. 64 x 1296 positions in another order than BLAS), 3e-2 bf16.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_full as mgf  # noqa: E402

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_GRAD, TOL_BF16 = 1e-4, 5e-4, 3e-2


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(mgf.PATH))


@pytest.fixture(scope="module")
def dbm():
    import deepbedmap_amd as d

    return d


@pytest.fixture(autouse=True)
def _reset_config(dbm):
    dbm.global_config.train = True
    dbm.global_config.enable_backprop = True
    dbm.global_config.ssim_window = "gaussian"
    dbm.global_config.cudnn_deterministic = True
    yield


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def copy_params(dst, src_params, persistent=None):
    for name, p in dst._tensors.items():
        if name in src_params:
            p.array = src_params[name]
        elif persistent is not None and name in persistent:
            p.array = np.asarray(persistent[name], dtype=np.float32)
    return dst


def grads_of(model):
    return {k: t.grad for k, t in model._tensors.items() if t.kind == 0}


DEV = 3.0    # the HIP path may be this many times further from the float64 oracle than the float32 oracle is
DEV_G = 5.0  # ... and for the generator's gradients of c3, than the float32 oracle's WORST tensor is (coherent flip noise)


def _close(got, ref64, ref32, rtol, atol=1e-6):
    """|got - float64 oracle| within rtol, or within DEV x the float32 oracle's own distance from it."""
    got, ref64, ref32 = (np.asarray(v, np.float64) for v in (got, ref64, ref32))
    tol = np.maximum(rtol * np.abs(ref64) + atol, DEV * np.abs(ref32 - ref64))
    return bool(np.all(np.abs(got - ref64) <= tol))


def _c3_models(dbm):
    og, od = mgf.models_c3()
    g = copy_params(dbm.GeneratorModel(num_residual_blocks=12, initialize=False), og.params)
    d = copy_params(dbm.DiscriminatorModel(initialize=False), od.params, od.persistent)
    d_opt = dbm.optimizers.Adam(alpha=mgf.ALPHA, eps=mgf.EPS).setup(d)
    g_opt = dbm.optimizers.Adam(alpha=mgf.ALPHA, eps=mgf.EPS).setup(g)
    return g, d, g_opt, d_opt


def _c3_check_d(d, got_d, gold):
    assert _close(got_d[0], gold["c3/d_step"][0], gold["c3/d_step_f32"][0], 2e-4), (got_d, gold["c3/d_step"])
    assert abs(got_d[1] - gold["c3/d_step"][1]) <= 2.0 / 128 + 1e-6, (got_d, gold["c3/d_step"])  # a logit near 0 may flip
    # One LeakyReLU slope that flips (a BatchNorm output within a rounding error of zero: which one depends on the
    # summation order, i.e. on the kernel's K split) moves ONE channel of that layer's beta / gamma gradient by a few per
    # cent and, through the BatchNorm backward passes below it, every earlier tensor by ~1e-3 -- the float32 oracle shows
    # the same on the layers where ITS flips sit (dev up to 6e-3).  Like the generator's gradients, the tensors are held
    # to a multiple of the WORST deviation the float32 oracle shows on any of them; everything above the flipped layer
    # stays at 1e-5 (conv_layer9, the linear layers), and the tight per-layer checks are test_gpu_ops / test_gpu_model.
    dev = gold["c3/gradD/dev"]
    worst = mgf.check_digest_dict(gold, "c3/gradD/", grads_of(d), max(TOL_GRAD, DEV_G * float(dev[:, 0].max())),
                                  max(TOL_GRAD, DEV_G * float(dev[:, 1].max())), floor=mgf.D_FLOOR, floors=mgf.ZERO_GRAD_FLOORS)
    assert worst[0] < 1.0, worst
    pers = {k: t.array for k, t in d._tensors.items() if t.kind == 1 and not k.endswith("/N")}
    worst = mgf.check_digest_dict(gold, "c3/persD/", pers, 1e-4, 1e-4, floor=mgf.G_FLOOR, dev_factor=DEV)
    assert worst[0] < 1.0, worst


def _c3_check_g(g, got_g, gold):
    # (the adversarial term sees D after its first Adam step, ~alpha * sign(gradient): rounding-noise gradients may flip)
    assert _close(got_g, gold["c3/g_step"], gold["c3/g_step_f32"], 5e-4, 1e-5), (got_g, gold["c3/g_step"])
    dev = gold["c3/gradG/dev"]
    worst = mgf.check_digest_dict(gold, "c3/gradG/", grads_of(g), max(TOL_GRAD, DEV_G * float(dev[:, 0].max())),
                                  max(TOL_GRAD, DEV_G * float(dev[:, 1].max())), floor=mgf.G_FLOOR)
    assert worst[0] < 1.0, worst


def test_config3_full_iteration_matches_oracle_fixture(dbm, gold):
    """BASELINE config 3: batch 64, 12 RRDB, D-step + G-step exactly as bench.py runs them (device-resident batch,
    prefetched G-step forward, cudnn_deterministic): forward, metrics, BatchNorm running statistics and EVERY gradient
    of both models against the float64 oracle (tolerances: module docstring of make_golden_full, Conditioning)."""
    g, d, g_opt, d_opt = _c3_models(dbm)
    batch = dbm.device_batch(mgf.arrays(64, 4200))
    with dbm.using_config("enable_backprop", False):
        y = g.forward(batch["X"], batch["W1"], batch["W2"], batch["W3"]).array.get()
    assert rel(y, gold["c3/g_forward"]) < max(TOL_FWD, DEV * float(gold["c3/g_forward_dev"]))
    got_d = dbm.train_eval_discriminator(batch, g, d, d_opt, prefetch_generator_forward=True)
    _c3_check_d(d, got_d, gold)
    got_g = dbm.train_eval_generator(batch, g, d, g_opt)
    _c3_check_g(g, got_g, gold)


def test_config3_sequential_path_gives_the_same_numbers(dbm, gold):
    """The same iteration without the prefetch and through host arrays (the reference's calling pattern), and with
    cudnn_deterministic = False (fp32 atomics over the K split of the weight gradients)."""
    for det in (True, False):
        with dbm.using_config("cudnn_deterministic", det):
            g, d, g_opt, d_opt = _c3_models(dbm)
            a = mgf.arrays(64, 4200)
            got_d = dbm.train_eval_discriminator(a, g, d, d_opt)
            _c3_check_d(d, got_d, gold)
            got_g = dbm.train_eval_generator(a, g, d, g_opt)
            _c3_check_g(g, got_g, gold)


def test_config3_generator_step_reference_init_tight(dbm, gold):
    """The G-step of config 3 with the reference's initialisation (c3lin: no slope hangs on a rounding error): loss, PSNR,
    SSIM and EVERY generator gradient at 5e-4 against the float32 oracle -- the persistent forward and backward trunk
    kernels at 192 workgroups, the tail, the deformable layers, the batched weight gradients with their ordered folds."""
    og, od = mgf.models_c3lin()
    g = copy_params(dbm.GeneratorModel(num_residual_blocks=12, initialize=False), og.params)
    d = copy_params(dbm.DiscriminatorModel(initialize=False), od.params, od.persistent)
    g_opt = dbm.optimizers.Adam(alpha=mgf.ALPHA, eps=mgf.EPS).setup(g)
    batch = dbm.device_batch(mgf.arrays(64, 6200))
    got = dbm.train_eval_generator(batch, g, d, g_opt)
    assert np.allclose(got, gold["c3lin/g_step"], rtol=2e-4, atol=1e-6), (got, gold["c3lin/g_step"])
    worst = mgf.check_digest_dict(gold, "c3lin/gradG/", grads_of(g), TOL_GRAD, TOL_GRAD)
    assert worst[0] < 1.0, worst


def test_config2_generator_only_l1_matches_oracle_fixture(dbm, gold):
    """BASELINE config 2: generator only, 16 RRDB, batch 32, pixel-L1 loss."""
    og = mgf.models_c2()
    g = copy_params(dbm.GeneratorModel(num_residual_blocks=16, initialize=False), og.params)
    a = mgf.arrays(32, 3100)
    t = mgf.target_c2()
    ins = [dbm.to_device(a[k]) for k in ("X", "W1", "W2", "W3")]
    y = g.forward(*ins).array.get()
    assert rel(y, gold["c2/g_forward"]) < TOL_FWD
    loss = float(np.abs(y - t).mean())
    assert abs(loss - float(gold["c2/loss"])) < 1e-5
    ref_y = gold["c2/g_forward"]
    gy = (np.sign(ref_y - t) / np.float32(ref_y.size)).astype(np.float32)  # the oracle's gy: sign() must not flip on noise
    g.cleargrads()
    g.backward(gy)
    worst = mgf.check_digest_dict(gold, "c2/gradG/", grads_of(g), TOL_GRAD, TOL_GRAD)
    assert worst[0] < 1.0, worst


def _c5_check(y, gold, tol):
    assert tuple(y.shape) == tuple(gold["c5/shape"])
    scale = float(gold["c5/stats"][1])
    c = y.shape[2] // 2 - mgf.C5_BLOCK // 2
    e_grid = np.abs(y[0, 0, ::mgf.C5_STRIDE, ::mgf.C5_STRIDE] - gold["c5/grid"]).max() / scale
    e_centre = np.abs(y[0, 0, c:c + mgf.C5_BLOCK, c:c + mgf.C5_BLOCK] - gold["c5/centre"]).max() / scale
    s, st = mgf.digest("c5/y", y)
    e_sample = np.abs(s - gold["c5/samples"]).max() / scale
    e_proj = np.abs(st[2:] - gold["c5/stats"][2:]).max() / gold["c5/stats"][0]
    assert max(e_grid, e_centre, e_sample, e_proj) < tol, (e_grid, e_centre, e_sample, e_proj)
    return max(e_grid, e_centre, e_sample)


def test_config5_full_crop_fp32_and_bf16_match_oracle_fixture(dbm, gold):
    """BASELINE config 5's unit of work: one interior 288 x 288 crop of the continent sweep -> 1144 x 1144
    (deepbedmap.py:706-728), fp32 against the oracle at 1e-4 and the bf16 sweep mode at 3e-2."""
    og = mgf.models_c5()
    g = copy_params(dbm.GeneratorModel(num_residual_blocks=12, initialize=False), og.params)
    a = mgf.arrays(1, 5100, h=288, w=288)
    ins = [dbm.to_device(a[k]) for k in ("X", "W1", "W2", "W3")]
    with dbm.using_config("enable_backprop", False):
        y32 = g.forward(*ins).array.get()
        with dbm.using_config("dtype", "bfloat16"):
            y16 = g.forward(*ins).array.get()
    _c5_check(y32, gold, TOL_FWD)
    _c5_check(y16, gold, TOL_BF16)
    # really the bf16 arithmetic -- in the TRUNK only since round 3 (DBM_BF16_FP32_LAYERS = 27: the layers on the signal path
    # keep fp32, DESIGN.md "bf16 at the data range"), whose residual branches enter through two 0.1 scalings: at the
    # reference's initialisation the two outputs differ in the seventh digit
    assert not np.array_equal(y16, y32) and np.abs(y16 - y32).max() / np.abs(y32).max() < 1e-3


def test_config5_sweep_slice_two_ranks_bf16(dbm, gold):
    """Config 5's loop on a slice of the continent that holds full-size tiles: 2 x 2 output tiles of 1000 x 1000 with the
    reference's 18-pixel halo (interior crops are 288 x 288; deepbedmap.py:689-741), grids resident in HBM, tiles dealt
    round-robin over two ranks with no collective, bf16 arithmetic; the fp32 resident sweep of the same area is the
    yardstick (3e-2) and the two-rank canvas must equal the one-rank canvas bitwise."""
    og = mgf.models_c5()
    g = copy_params(dbm.GeneratorModel(num_residual_blocks=12, initialize=False), og.params)
    H, W = 500, 500
    r = np.random.RandomState(9)
    X = r.rand(1, 1, H, W).astype(np.float32)
    W1 = r.rand(1, 1, 10 * H, 10 * W).astype(np.float32)
    W2 = r.rand(1, 2, 2 * H, 2 * W).astype(np.float32)
    W3 = r.rand(1, 1, H, W).astype(np.float32)
    S = dbm.Shape
    final = S(y=4 * H, x=4 * W)
    grids = [dbm.to_device(v) for v in (X, W1, W2, W3)]
    kw = dict(final_shape=final, ary_shape=S(y=1000, x=1000), stride=S(y=1000, x=1000), xtrapad=S(y=18, x=18))
    shapes = {(y1 - y0, x1 - x0) for y0, y1, x0, x1 in
              (dbm.crop_bounds(s, final, kw["ary_shape"], kw["xtrapad"]) for s in dbm.tile_steps(final, kw["stride"]))}
    assert shapes == {(269, 269)}  # edge tiles of a 2 x 2 area (interior tiles of the continent: 288 x 288)
    y32 = dbm.predict_tiled_resident(g, *grids, **kw)
    y16 = dbm.predict_tiled_resident(g, *grids, dtype="bfloat16", **kw)
    parts = [dbm.predict_tiled_resident(g, *grids, dtype="bfloat16", rank=k, world=2, **kw) for k in range(2)]
    m = ~np.isnan(y32)
    frame = (18 + 1) * 4
    assert m[:, frame:-frame, frame:-frame].all() and not m[:, :frame].any()
    assert np.array_equal(np.isnan(y16), ~m)
    err = np.abs(y16[m] - y32[m]).max() / np.abs(y32[m]).max()
    assert 0 < err < TOL_BF16, err  # (bf16 in the trunk only: see test_config5_full_crop_fp32_and_bf16_match_oracle_fixture)
    assert np.array_equal(np.nan_to_num(dbm.merge_ranks(parts), nan=-1.0), np.nan_to_num(y16, nan=-1.0))

@pytest.mark.parametrize("form", ["helpers_in_every_pass", "no_helpers", "tiles_of_32_positions"])
def test_config3_other_forms_of_the_trunk_forward_match_the_fixture(form):
    """The trunk forward kernel's other forms (DBM_TRUNK_HELPER / DBM_TRUNK_TP, read once per process) against the same
    batch-64 oracle fixtures: the config 3 iteration and the tight generator-step check, in a process of their own."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if not k.startswith("DBM_TRUNK_")}
    env.update({"helpers_in_every_pass": {"DBM_TRUNK_HELPER": "3"}, "no_helpers": {"DBM_TRUNK_HELPER": "0"},
                "tiles_of_32_positions": {"DBM_TRUNK_TP": "32"}}[form])
    here = os.path.abspath(__file__)
    res = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", "-k",
                          "config3_full_iteration or config3_generator_step_reference_init"],
                         env=env, cwd=os.path.dirname(os.path.dirname(here)), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "2 passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("form", ["post_residual_in_fp32_igemm", "pre_and_post_residual_in_fp32_igemm", "deformable_layer_gathers_from_memory"])
def test_config5_other_forms_of_the_sweep_tail_match_the_fixture(form):
    """The bf16 sweep's switchable layer forms (read once per process) against the same 288 x 288 crop fixture: bit 32 of
    DBM_BF16_FP32_LAYERS (the post-residual convolution as an fp32 igemm launch between layout conversions instead of split-bf16 on the
    channels-last planes), DBM_POST_X3=0 / DBM_PRE_X3=0 (both of them; round 5's A/B switches) and DBM_DEFORM_X3_WINDOW=0 (round 6: the
    64 -> 64 deformable layer's sampler gathers every corner from memory instead of reading an LDS window)."""
    import subprocess

    env = dict(os.environ)
    env.update({"post_residual_in_fp32_igemm": {"DBM_BF16_FP32_LAYERS": "59"},
                "pre_and_post_residual_in_fp32_igemm": {"DBM_POST_X3": "0", "DBM_PRE_X3": "0"},
                "deformable_layer_gathers_from_memory": {"DBM_DEFORM_X3_WINDOW": "0"}}[form])
    here = os.path.abspath(__file__)
    res = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", "-k",
                          "config5_full_crop_fp32_and_bf16_match_oracle_fixture"],
                         env=env, cwd=os.path.dirname(os.path.dirname(here)), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "1 passed" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("cpb", [1, 8])
def test_reduced_continent_every_crop_kind_bf16(dbm, cpb):
    """Config 5 on a 3 x 3-tile continent (3000 x 3000 output pixels at the reference's stride 1000 and 18-pixel halo,
    deepbedmap.py:689-741): ONE interior crop of 288 x 288, FOUR edge crops (269 x 288 / 288 x 269) and FOUR corner crops
    (269 x 269) -- every crop kind of the real sweep -- bf16, grids resident in HBM, one and eight crops per forward, against the
    per-tile loop (predict_tiled: host crops, one forward each, the reference's own loop form)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dem_model import dem_generator

    g = dem_generator(dbm, seed=909)   # activations at the data's magnitude (metres)
    H = W = 750
    r = np.random.RandomState(11)
    X = r.uniform(-2000, 2000, (1, 1, H, W)).astype(np.float32)
    W1 = r.uniform(-100, 4000, (1, 1, 10 * H, 10 * W)).astype(np.float32)
    W2 = r.uniform(-10, 1000, (1, 2, 2 * H, 2 * W)).astype(np.float32)
    W3 = r.uniform(0, 500, (1, 1, H, W)).astype(np.float32)
    S = dbm.Shape
    final = S(y=4 * H, x=4 * W)
    kw = dict(final_shape=final, ary_shape=S(y=1000, x=1000), stride=S(y=1000, x=1000), xtrapad=S(y=18, x=18))
    groups = dbm.group_tiles_by_crop_shape(final, kw["ary_shape"], kw["stride"], kw["xtrapad"])
    assert {k: len(v) for k, v in groups.items()} == {(269, 269): 4, (269, 288): 2, (288, 269): 2, (288, 288): 1}
    W1c, W2c, W3c = dbm.clip_inputs(W1, W2, W3)   # deepbedmap.py:663-665 (host arrays: new arrays)
    ref = dbm.predict_tiled(g, X, W1c, W2c, W3c, dtype="bfloat16", **kw)
    got = dbm.predict_tiled_resident(g, X, W1, W2, W3, dtype="bfloat16", clip=True, crops_per_batch=cpb, **kw)
    m = ~np.isnan(ref)
    frame = (18 + 1) * 4
    assert np.array_equal(np.isnan(got), ~m) and m[:, frame:-frame, frame:-frame].all() and not m[:, :frame].any()
    scale = float(np.abs(ref[m]).max())
    assert scale > 100.0   # metres, not U[0, 1)
    if cpb == 1:
        assert np.array_equal(got[m], ref[m])   # the same launches per crop: bit for bit
    else:   # eight crops per forward: launch splits depend on the batch size -- another summation order, and in this mode a
        # last-bit difference in fp32 can flip a bf16 rounding of the trunk's activations: bf16-level agreement (measured 1.3e-4)
        assert np.abs(got[m] - ref[m]).max() / scale < 1e-3


def test_bench_sweep_continent_entry_on_a_reduced_area(dbm):
    """`bench.py --sweep-continent` (the config-5 entry a SCALE run uses), on a 1/6-scale area so that it takes seconds: one JSON
    line below 4 KB with s_per_continent, ranks and a canvas check."""
    import json
    import subprocess

    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sweep-continent", "--sweep-scale", "6"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["ranks"] == 1 and d["n_gpus"] == 1 and d["dtype"] == "bf16" and d["higher_is_better"] is False
    assert d["s_per_continent"] == d["value"] > 0 and d["canvas_check_ok"] is True
    assert "750 x 916" not in d["config"]["workload"] and "3000 x 3664" in d["config"]["workload"]
