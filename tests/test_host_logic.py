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
    rows = bench.shape_table(in_step, serial, ["per_layer_conv_family", "wgrad_kernel", "trunk_fused_kernel"])
    assert [r["shape"] for r in rows] == ["trunk_fwd_36rdb_n64_keep", "c64>64_k9_36x36"]  # sorted by standalone time
    conv = rows[1]
    assert conv["launches"] == 2 and conv["workgroups"] == 2592 and abs(conv["ms"] - 0.22) < 1e-9 and abs(conv["ms_standalone"] - 0.17) < 1e-9
    assert abs(conv["tflops_standalone"] - 2 * 6.1 / 0.17) < 1e-2
    assert abs(conv["frac_mfma_standalone"] - conv["tflops_standalone"] / bench.PEAK_FP32_MFMA_TFLOPS) < 1e-3
    assert abs(conv["algorithmic_gbps_standalone"] - 2 * 4.0e7 / 0.17e-3 / 1e9) < 1.0
    assert rows[0]["kernel"] == "trunk_fused_kernel"


def _canned_bench_inputs():
    fam = [{"key": "per_layer_conv_family", "label": "x" * 150, "ms_per_step": 5.61234567, "launches_per_step": 72, "flop": 134.8e9, "bytes": 1.55e9,
            "standalone_ms_per_step": 2.94123},
           {"key": "wgrad_kernel", "label": "y" * 90, "ms_per_step": 3.9, "launches_per_step": 13, "flop": 138.3e9, "bytes": 0.9e9, "standalone_ms_per_step": 2.26},
           {"key": "trunk_fused_kernel<retained>", "label": "", "ms_per_step": 1.286, "launches_per_step": 1, "flop": 89.44e9, "bytes": 1.8e8,
            "standalone_ms_per_step": 1.257},
           {"key": "trunk_fused_bwd_kernel", "label": "", "ms_per_step": 1.551, "launches_per_step": 1, "flop": 89.44e9, "bytes": 2.0e8,
            "standalone_ms_per_step": 1.536},
           {"key": "trunk_fused_kernel<helper>", "label": "", "ms_per_step": 1.019, "launches_per_step": 1, "flop": 89.44e9, "bytes": 1.8e8,
            "standalone_ms_per_step": 1.047}]
    mode = lambda ms: {"ms_per_crop": ms, "tflops": 1707.0 / ms, "frac_of_mfma_peak": 0.1191, "mfma_peak_tflops": 2500.0, "compulsory_gbs": 12.3,
                       "s_per_continent_one_gpu": 0.396 * ms, "bracketed_ms": ms * 1.1, "per_shape_standalone": [{"shape": "cl16_c64>32_286x286_n1"}] * 40}
    sweep = {"crop": [288, 288], "output": [1144, 1144], "algorithmic_tflop_per_crop": 1.7071, "compulsory_bytes_per_crop": 5.7e7, "crops_timed": 5,
             "fp32": mode(19.9), "bf16": dict(mode(5.73), batch8={"ms_per_crop": 5.36, "tflops": 318.0, "frac_of_mfma_peak": 0.127,
                                                                 "s_per_continent_one_gpu": 2.12})}
    cpu = {"value": 1.0412345, "unit": "tiles/s", "cores": 64, "kind": "port", "host_cpus": 256, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
           "blas_threads": 64, "measured_s": [15.4, 15.2], "sample": "2 full iteration(s) ..." + "z" * 120, "wall_s": 61.0,
           "torch_cpu": {"value": 13.6, "unit": "tiles/s", "threads": 64, "sample": "torch 2.10 CPU fp32 ..." + "w" * 80}}
    config = {"workload": "full ESRGAN training iteration (D-step + G-step, fwd+bwd+Adam), 12 RRDB, 11x11 -> 36x36 tiles, fp32",
              "batch_per_gpu": 64, "global_batch": 64, "parallelism": "dp1", "generator_forwards_per_iteration": 2,
              "g_step_forward_prefetched_under_d_step": True, "cudnn_deterministic": True, "fused_iteration_call": True,
              "metrics_read_back": "once per run", "sync_batch_stats": False}
    return dict(tiles=64 * 200, dt=1.6288, steps=200, warmup=10, world=1, batch=64, ev_ms=1628.1, config=config, fam=fam,
                traffic={"hbm_bytes_per_launch": 57.0e6, "source": "static: profiles/r4/traffic_pmc.json (rocprofv3 --pmc, separate passes; not this run)"},
                sweep=sweep, shared={"ms_per_step": 7.46, "tiles_per_s": 8579.1, "steps": 40, "note": "n" * 500}, cpu=cpu,
                tables_path="/root/repo/bench_tables.json")


def test_bench_line_is_short_and_complete():
    """The driver keeps an 8 KB tail of stdout: the ONE JSON line must stay below 4 KB and still carry everything the contract
    names (round 3's 30 KB line was unreadable to it).  Built from canned measurements -- no GPU."""
    import json

    import bench

    out = bench.compose_line(env={"DBM_IGEMM_WAVES": "8"}, **_canned_bench_inputs())
    line = bench.fit_line(out)
    assert len(line.encode()) < bench.MAX_LINE_BYTES == 4096, len(line)
    assert "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "dropped_for_length" not in d
    assert d["config"]["workload"] and d["config"]["env"] == {"DBM_IGEMM_WAVES": "8"}
    assert abs(d["value"] - 64 * 200 / 1.6288) < 0.01 and abs(d["ms_per_step"] - 8.144) < 1e-3
    rf = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_standalone", "frac_step", "traffic", "traffic_over_algorithmic_bytes",
              "rrdb_forward", "other_kernels"):
        assert k in rf, k
    assert rf["kernel"] == "per_layer_conv_family" and rf["bound"] == "mfma" and rf["peak"] == bench.PEAK_FP32_MFMA_TFLOPS
    assert abs(rf["achieved"] - 134.8 / 5.61234567) < 1e-2 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["frac_standalone"] - 134.8 / 2.94123 / 157.3) < 1e-3
    assert abs(rf["frac_step"] - 8.43 * 64 / 8.144 / 157.3) < 1e-3
    assert rf["rrdb_forward"]["form"] == "trunk_fused_kernel<helper>" and abs(rf["rrdb_forward"]["frac_standalone"] - 89.44 / 1.047 / 157.3) < 1e-3
    assert {o["kernel"] for o in rf["other_kernels"]} == {"wgrad_kernel", "trunk_fused_kernel<retained>", "trunk_fused_bwd_kernel", "trunk_fused_kernel<helper>"}
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 64 and cb["torch_cpu"]["value"] == 13.6 and "sample" in cb
    assert d["extras"]["sweep"]["bf16"]["ms_per_crop"] == 5.73 and d["extras"]["sweep"]["fp32"]["ms_per_crop"] == 19.9
    assert d["extras"]["sweep"]["bf16"]["batch8_ms_per_crop"] == 5.36
    assert d["tables"] == "bench_tables.json"
    # no table rides in the line
    assert "per_shape" not in line and "per_shape_standalone" not in line


def test_bench_line_sheds_detail_rather_than_grow():
    import json

    import bench

    kw = _canned_bench_inputs()
    out = bench.compose_line(env={f"DBM_SWITCH_{i}": "v" * 200 for i in range(40)}, **kw)
    line = bench.fit_line(out)
    d = json.loads(line)
    assert len(line.encode()) < 4096 and "config.env" in d["dropped_for_length"]
    assert sorted(d["config"]["env"]) == sorted(f"DBM_SWITCH_{i}" for i in range(40))  # the names survive
    for k in ("value", "ms_per_step", "roofline", "cpu_baseline"):
        assert k in d


def test_bench_refuses_work_skipping_switches_and_prices_split_bf16_correctly():
    import bench

    bench.refuse_work_skipping_env({"DBM_IGEMM_WAVES": "8", "PATH": "/bin"})
    for name in ("DBM_ABL_SKIP", "DBM_NO_WGRAD", "DBM_TFB_ABL", "DBM_CL16_ABL", "DBM_ABL_NOPACK"):
        with pytest.raises(SystemExit):
            bench.refuse_work_skipping_env({name: "1"})
    assert bench.dbm_env({"DBM_B": "2", "DBM_A": "1", "HOME": "/"}) == {"DBM_A": "1", "DBM_B": "2"}
    # three bf16 MFMAs per product in the split-bf16 kernels: a third of the bf16 roof (round 3 divided them by the fp32 peak: 2.13)
    assert bench.sweep_roof("bf16", "x3_c64>64_1144x1144_n1u") == ("bf16_mfma/3", 2500.0 / 3)
    assert bench.sweep_roof("bf16", "deform64x3_1144x1144_n1")[1] == 2500.0 / 3
    assert bench.sweep_roof("bf16", "cl16_c64>32_286x286_n1")[1] == 2500.0
    assert bench.sweep_roof("bf16", "deform1_1144x1144_n1")[1] == bench.PEAK_FP32_MFMA_TFLOPS
    assert bench.sweep_roof("fp32", "x3_c64>64_1144x1144_n1u")[1] == bench.PEAK_FP32_MFMA_TFLOPS


def test_committed_bench_tables_belong_to_the_line_beside_them():
    """`profiles/<round>/<prefix>_bench_tables.json` must be the tables OF the run whose line is
    `<prefix>_bench.json` (round 4 committed tables a later --pmc pass had overwritten).  For every such pair from round 5 on:
    every kernel family's standalone total in the tables equals standalone_avg_launch_us x launches (dominant family) /
    ms_standalone (the others) of the line, within 5 %."""
    import glob
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = []
    for rnd in sorted(glob.glob(os.path.join(root, "profiles", "r[5-9]"))):
        for tp in sorted(glob.glob(os.path.join(rnd, "*_bench_tables.json"))):
            lp = tp[: -len("_tables.json")] + ".json"
            if os.path.exists(lp):
                pairs.append((tp, lp))
    for tp, lp in pairs:
        tables = json.load(open(tp))
        line = json.loads(open(lp).read().strip().splitlines()[-1])
        fam = {f["key"]: f for f in tables["families"]}
        roof = line["roofline"]
        want = {roof["kernel"]: 1e-3 * roof["standalone_avg_launch_us"] * roof["launches_per_step"]}
        for o in roof.get("other_kernels", []):
            want[o["kernel"]] = o["ms_standalone"]
        assert want, lp
        for k, ms in want.items():
            got = fam[k]["standalone_ms_per_step"]
            assert abs(got - ms) <= 0.05 * ms + 2e-3, (os.path.basename(tp), k, got, ms)
