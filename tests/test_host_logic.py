"""CPU tests of the host-side logic that needs no GPU: iterator semantics (srgan_train.py:132-166, 1286-1288,
1311-1313), tiling index arithmetic (deepbedmap.py:689-741), Chainer parameter ordering, config flags."""
import numpy as np
import pytest

import deepbedmap_amd as dbm
from deepbedmap_amd.srgan import _chainer_order
from oracle import model as omodel


def test_serial_iterator_full_batches_and_epochs():
    # 3826 training tiles at batch 128 -> 30 iterations per epoch, always full batches
    data = {"X": np.arange(3826)}
    it = dbm.SerialIterator(data, batch_size=128, repeat=True, shuffle=True, seed=42)
    n_iter, seen = 0, []
    while it.epoch == 0:
        idx = it.next()
        assert len(idx) == 128
        seen.append(idx)
        n_iter += 1
    assert n_iter == 30
    first_epoch = np.concatenate(seen)[:3826]
    assert np.array_equal(np.sort(first_epoch), np.arange(3826))  # every tile exactly once before wrapping
    # dev iterator: 202 tiles, no shuffle -> 2 iterations
    dev = dbm.SerialIterator({"X": np.arange(202)}, batch_size=128, repeat=True, shuffle=False)
    k = 0
    while dev.epoch == 0:
        dev.next()
        k += 1
    assert k == 2
    batch = dbm.concat_examples({"X": np.arange(10) * 2.0}, np.array([3, 1]))
    assert np.array_equal(batch["X"], [6.0, 2.0])


def test_tiling_covers_everything_but_the_frame():
    S = dbm.Shape
    final, ary, stride, pad = S(y=18000, x=22000), S(y=1000, x=1000), S(y=1000, x=1000), S(y=18, x=18)
    steps = dbm.tile_steps(final, stride)
    assert len(steps) == 396  # 18 x 22 crops (deepbedmap.py:700-703)
    cover = np.zeros((final.y // 4, final.x // 4), np.int32)  # in units of 4x4 output pixels
    sizes = set()
    for st in steps:
        y0, y1, x0, x1 = dbm.crop_bounds(st, final, ary, pad)
        sizes.add((y1 - y0, x1 - x0))
        cover[y0 + pad.y + 1:y1 - pad.y - 1, x0 + pad.x + 1:x1 - pad.x - 1] += 1
    assert (288, 288) in sizes  # interior crops: 250 + 2*18 + 2 low-resolution pixels
    frame = pad.y + 1
    assert (cover[frame:-frame, frame:-frame] == 1).all()  # written exactly once
    assert cover[:frame].sum() == 0 and cover[:, :frame].sum() == 0  # the outer 76-px frame stays NaN


def test_chainer_param_order_matches_the_doctest_indices():
    names = list(omodel.generator_param_shapes(12))
    order = _chainer_order(names)
    assert order[8] == "input_block/conv_on_W1/W"  # srgan_train.py:1203
    assert order[:4] == ["final_conv_layer1/deform_conv/W", "final_conv_layer1/deform_conv/b",
                         "final_conv_layer1/offset_conv/W", "final_conv_layer1/offset_conv/b"]
    dorder = _chainer_order(list(omodel.discriminator_param_shapes()))
    assert dorder[-3] == "linear_1/b"  # srgan_train.py:1113
    assert order == omodel.chainer_param_order(names)


def test_using_config_restores_flags():
    assert dbm.global_config.enable_backprop is True
    with dbm.using_config("enable_backprop", False):
        assert dbm.global_config.enable_backprop is False
        with dbm.using_config("train", False):
            assert dbm.global_config.train is False
    assert dbm.global_config.enable_backprop is True


def test_residual_scaling_and_blocks_are_plain_attributes():
    # deepbedmap.py:402-405 / srgan_train.py:1577-1578 read them back; they are not serialized
    shapes = omodel.generator_param_shapes(3)
    assert not any("residual_scaling" in k or "num_residual_blocks" in k for k in shapes)


def test_get_train_dev_iterators_split_semantics():
    """srgan_train.py:132-166 / chainer.datasets.split_dataset_random: one seeded permutation, disjoint and complete."""
    import deepbedmap_amd.training as tr

    n = 40
    ds = {"X": np.arange(n, dtype=np.float32).reshape(n, 1, 1, 1), "Y": 10 * np.arange(n, dtype=np.float32).reshape(n, 1, 1, 1)}
    train_iter, n_train, dev_iter, n_dev = tr.get_train_dev_iterators(ds, first_size=int(n * 0.95), batch_size=8, seed=42)
    assert (n_train, n_dev) == (38, 2)
    order = np.random.RandomState(42).permutation(n)
    np.testing.assert_array_equal(train_iter.dataset["X"].ravel(), order[:38].astype(np.float32))
    np.testing.assert_array_equal(dev_iter.dataset["X"].ravel(), order[38:].astype(np.float32))
    np.testing.assert_array_equal(train_iter.dataset["Y"].ravel(), 10 * train_iter.dataset["X"].ravel())  # rows stay paired
    assert train_iter.shuffle and not dev_iter.shuffle and train_iter.repeat and dev_iter.repeat
    # the dev iterator walks its two tiles in order, wrapping around to fill the batch of 8
    np.testing.assert_array_equal(dev_iter.next()[:2], [0, 1])
    with pytest.raises(ValueError):
        tr.split_dataset_random(ds, first_size=n + 1, seed=0)


def test_continent_tiles_grouped_by_crop_shape():
    """deepbedmap.py:689-741 at the continent's size: 396 tiles; the resident sweep batches crops of equal shape -- 320 interior
    crops of 288 x 288 low-resolution pixels, 40 + 32 edge crops, 4 corners; two ranks split every group without overlap."""
    from deepbedmap_amd.inference import Shape, group_tiles_by_crop_shape

    final, ary, pad = Shape(y=18000, x=22000), Shape(y=1000, x=1000), Shape(y=18, x=18)
    g = group_tiles_by_crop_shape(final, ary, ary, pad)
    assert {k: len(v) for k, v in g.items()} == {(288, 288): 320, (269, 288): 40, (288, 269): 32, (269, 269): 4}
    for (h, w), tiles in g.items():
        assert all(y1 - y0 == h and x1 - x0 == w for y0, y1, x0, x1 in tiles)
    parts = [group_tiles_by_crop_shape(final, ary, ary, pad, rank=r, world=2) for r in range(2)]
    both = [t for p in parts for tiles in p.values() for t in tiles]
    assert len(both) == 396 and len(set(both)) == 396
    assert sorted(both) == sorted(t for tiles in g.values() for t in tiles)


def test_bench_shape_table_aggregates_profiler_records():
    """bench.py's per-shape roofline table: records of an in-step and of a serialised step (Context.profile_records()) grouped by
    (family, tag, flops, bytes); standalone TFLOP/s, fraction of the fp32 MFMA roof and the HBM rate the algorithmic bytes imply."""
    import bench

    rec = lambda fam, tag, fl, by, ms, wgs: {"family": fam, "tag": tag, "flops": fl, "bytes": by, "ms": ms, "wgs": wgs}
    in_step = [rec(0, "c64>64_k9_36x36", 6.1e9, 4.0e7, 0.12, 2592), rec(0, "c64>64_k9_36x36", 6.1e9, 4.0e7, 0.10, 2592),
               rec(2, "trunk_fwd_36rdb_n64_keep", 8.94e10, 1.8e8, 1.3, 192)]
    serial = [rec(0, "c64>64_k9_36x36", 6.1e9, 4.0e7, 0.08, 2592), rec(0, "c64>64_k9_36x36", 6.1e9, 4.0e7, 0.09, 2592),
              rec(2, "trunk_fwd_36rdb_n64_keep", 8.94e10, 1.8e8, 1.25, 192)]
    rows = bench.shape_table(in_step, serial, ["igemm_conv_kernel", "wgrad_kernel", "trunk_fused_kernel"])
    assert [r["shape"] for r in rows] == ["trunk_fwd_36rdb_n64_keep", "c64>64_k9_36x36"]  # sorted by standalone time
    conv = rows[1]
    assert conv["launches"] == 2 and conv["workgroups"] == 2592 and abs(conv["ms"] - 0.22) < 1e-9 and abs(conv["ms_standalone"] - 0.17) < 1e-9
    assert abs(conv["tflops_standalone"] - 2 * 6.1 / 0.17) < 1e-2
    assert abs(conv["frac_mfma_standalone"] - conv["tflops_standalone"] / bench.PEAK_FP32_MFMA_TFLOPS) < 1e-3
    assert abs(conv["algorithmic_gbps_standalone"] - 2 * 4.0e7 / 0.17e-3 / 1e9) < 1.0
    assert rows[0]["kernel"] == "trunk_fused_kernel"
