#!/usr/bin/env python3
"""Headline benchmark: DEM tiles/sec of the full ESRGAN training iteration (D-step + G-step: generator and
discriminator forward + backward + Adam) at batch 64 per GPU, fp32, 12 RRDB -- BASELINE.json `metric`,
config "1xMI355X full ESRGAN ... batch 64, fp32" (the reference has no VGG/perceptual branch: SURVEY.md section 0).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One rank per GPU; per-GPU work is fixed (weak scaling): rank r trains on its own 64 synthetic tiles
(np.random.RandomState(42 + r).rand, the reference's fixture recipe, srgan_train.py:1101-1105) and gradients are
summed with one RCCL all-reduce per optimizer step.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def _traffic_json():  # tools/pmc_traffic.sh (separate --pmc passes); the newest round's file
    for r in ("r6", "r5", "r4", "r3", "r2", "r1"):
        p = os.path.join(ROOT, "profiles", r, "traffic_pmc.json")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r1", "traffic_pmc.json")


TRAFFIC_JSON = _traffic_json()
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA (the sweep's arithmetic, BASELINE config 5)
PEAK_HBM_GBS = 8000.0
# split-bf16 kernels (conv_cl16x3, deform_conv64_x3: hi*hi + hi*lo + lo*hi) issue THREE bf16 MFMAs per algorithmic product
PEAK_SPLIT_BF16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
MAX_LINE_BYTES = 4096   # the driver keeps an 8 KB tail of stdout: the result line stays far below it (tests/test_host_logic.py)
TABLES_PATH = os.path.join(ROOT, "bench_tables.json")   # the per-shape tables of the run (also echoed on stderr)
# environment switches of libdbm_measure.so (built by tools/build_measure.sh with -DDBM_MEASURE) that SKIP work: a run with
# one of them set is not a measurement of the iteration and bench.py refuses it
WORK_SKIPPING_ENV = ("DBM_ABL_SKIP", "DBM_NO_WGRAD", "DBM_TFB_ABL", "DBM_CL16_ABL", "DBM_ABL_NOPACK", "DBM_ITER_ABL", "DBM_LIB")
G_FWD_MAC_PER_TILE = 845360064  # generator forward, 12 RRDB, one 11x11 -> 36x36 tile (81 trunk pixels)
BATCH_PER_GPU = 64
N_RRDB = 12
GFLOP_PER_TILE = 8.43  # 2 G-fwd + 3 D-fwd + G-bwd + 2 D-bwd ~ 4 G_f + 7 D_f at 12 RRDB


def baseline_metric():
    """BASELINE.json's metric string (the file ships with the repository)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "DEM tiles/sec (fwd+bwd, gen+disc) at batch 64, 1/2/4/8 MI355X"


def synthetic_batch(n, seed):
    r = lambda *s: np.random.RandomState(seed=seed).rand(*s).astype(np.float32)  # noqa: E731
    return {"X": r(n, 1, 11, 11), "W1": r(n, 1, 110, 110), "W2": r(n, 2, 22, 22), "W3": r(n, 1, 11, 11),
            "Y": r(n, 1, 36, 36)}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(threads=None, batch=16):
    """The oracle (NumPy restatement of the Chainer CPU path: im2col + BLAS sgemm) on this box's host cores, on a BOUNDED
    sample of the benchmark's workload: full iterations (D-step + G-step: forward, backward, Adam; 12 RRDB) at batch 16 --
    a quarter of the benchmark's 64 tiles, the same per-tile arithmetic -- after a warm-up iteration at batch 4; `value` is the
    median of THREE measured iterations (16-25 s each on the round-4 boxes), and the oracle's generator forward alone is timed at
    N = 1 (BASELINE configs[0]; median of 10) and N = 64 (one pass after a warm-up) -- `g_forward_n1_ms`, `g_forward_n64_ms`.  Beside it
    the same iteration in torch-CPU fp32 (oneDNN convolutions, autograd) at the benchmark's batch 64, a strong-CPU yardstick.
    BLAS / torch threads are pinned and reported."""
    import statistics

    from oracle import model as omodel
    from oracle import train as otrain

    ncpu = os.cpu_count() or 1
    threads = threads or min(ncpu, 64)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=threads)
    except Exception:
        limit = None

    def port_step(n):
        arrays = synthetic_batch(n, 42)
        og = omodel.GeneratorModel(num_residual_blocks=N_RRDB, seed=1)
        od = omodel.DiscriminatorModel(seed=2)
        g_opt = otrain.Adam(og.params, alpha=1.6e-4)
        d_opt = otrain.Adam(od.params, alpha=1.6e-4)
        t0 = time.perf_counter()
        otrain.train_eval_discriminator(arrays, og, od, d_opt)
        otrain.train_eval_generator(arrays, og, od, g_opt)
        return time.perf_counter() - t0

    def port_forward(n, reps):
        """BASELINE configs[0] / BASELINE.md section 3 (a), (b): the generator forward alone (srgan_train.py:437-447, deepbedmap.py:420-421)."""
        arrays = synthetic_batch(n, 42)
        og = omodel.GeneratorModel(num_residual_blocks=N_RRDB, seed=1)
        og.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"])   # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            og.forward(arrays["X"], arrays["W1"], arrays["W2"], arrays["W3"])
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    t_all = time.perf_counter()
    tw = port_step(4)  # warm-up: BLAS thread pool, page faults of the im2col buffers
    ts = [port_step(batch) for _ in range(3)]   # (three samples, 16-25 s each on the round-4 boxes: the median is not one of two)
    tN = statistics.median(ts)
    f1, f64 = port_forward(1, 10), port_forward(BATCH_PER_GPU, 1)   # (N = 64: one warm-up + ONE timed pass, 13-14 s each on the round-5 box)
    out = {"value": batch / tN, "unit": "tiles/s", "cores": threads, "kind": "port",
           "host_cpus": ncpu, "cpu_model": _cpu_model(), "blas_threads": threads,
           "measured_s": [round(t, 3) for t in ts], "warmup_batch4_s": round(tw, 3),
           "g_forward_n1_ms": round(1e3 * f1, 2), "g_forward_n64_ms": round(1e3 * f64, 1),
           "sample": f"{len(ts)} full iterations (D+G step, fwd+bwd+Adam, 12 RRDB) of the NumPy/BLAS oracle at batch {batch} "
                     f"(median {tN:.1f} s) after a batch-4 warm-up; g_forward_*: generator forward alone, N = 1 (median of 10) / N = 64 (one pass)"}
    try:  # torch-CPU (oneDNN) fp32, the same iteration at the benchmark's batch
        import torch

        from oracle import torch_ref as tr

        torch.set_num_threads(threads)
        og = omodel.GeneratorModel(num_residual_blocks=N_RRDB, seed=1)
        od = omodel.DiscriminatorModel(seed=2)
        Pg, Pd = tr.tp(og.params, torch.float32), tr.tp(od.params, torch.float32)
        Sd = {k: torch.tensor(np.asarray(v, np.float32)) for k, v in od.persistent.items() if not k.endswith("/N")}
        nb = BATCH_PER_GPU
        arrays = {k: torch.tensor(v) for k, v in synthetic_batch(nb, 42).items()}

        def torch_step():
            t0 = time.perf_counter()
            tr.training_iteration(Pg, Pd, Sd, arrays, n_blocks=N_RRDB)
            return time.perf_counter() - t0

        torch_step()
        tt = statistics.median(torch_step() for _ in range(3))
        out["torch_cpu"] = {"value": nb / tt, "unit": "tiles/s", "threads": threads,
                            "sample": f"torch {torch.__version__} CPU fp32 (oneDNN, autograd), batch {nb}, 1 warm-up + median of 3 ({tt:.2f} s)"}
    except Exception as e:  # pragma: no cover
        out["torch_cpu"] = {"error": repr(e)[:200]}
    if limit is not None:
        limit.restore_original_limits() if hasattr(limit, "restore_original_limits") else None
    out["wall_s"] = round(time.perf_counter() - t_all, 1)
    return out


def shape_table(recs, recs_s, keys):
    """Per launch shape (family, tag): launches per step, in-step and standalone time, TFLOP/s and fraction of the fp32 MFMA
    roof (standalone), algorithmic bytes per launch and the GB/s they imply.  recs / recs_s: Context.profile_records() of an
    in-step and of a serialised step (same launches, same order)."""
    rows = {}
    for src, col in ((recs, "ms"), (recs_s, "ms_standalone")):
        for r in src:
            row = rows.setdefault((r["family"], r["tag"], r["flops"], r["bytes"]),
                                  {"kernel": keys[r["family"]], "shape": r["tag"], "workgroups": r["wgs"], "launches": 0, "ms": 0.0, "ms_standalone": 0.0,
                                   "gflop_per_launch": r["flops"] / 1e9, "algorithmic_bytes_per_launch": r["bytes"]})
            row[col] += r["ms"]
            if col == "ms":
                row["launches"] += 1
    out = []
    for row in rows.values():
        n, ms = max(row["launches"], 1), row["ms_standalone"]
        row["avg_us_standalone"] = 1e3 * ms / n
        row["tflops_standalone"] = row["gflop_per_launch"] * n / ms if ms > 0 else 0.0
        row["frac_mfma_standalone"] = row["tflops_standalone"] / PEAK_FP32_MFMA_TFLOPS
        row["algorithmic_gbps_standalone"] = row["algorithmic_bytes_per_launch"] * n / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        row["flop_per_byte"] = row["gflop_per_launch"] * 1e9 / max(row["algorithmic_bytes_per_launch"], 1.0)
        out.append({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()})
    return sorted(out, key=lambda r: -r["ms_standalone"])


def sweep_roof(mode, tag):
    """The MFMA roof a bracketed launch of the sweep is priced against.  fp32 mode: the fp32 MFMA peak.  bf16 mode: the bf16
    peak for the bf16 trunk layers (`cl16_...` = conv_cl16_kernel); a THIRD of it for the split-bf16 launches (`x3_...` =
    conv_cl16x3_kernel, `deform64x3_...` = deform_conv64_x3_kernel: hi*hi + hi*lo + lo*hi, three bf16 MFMAs per algorithmic
    product); the fp32 peak for whatever still multiplies in fp32 (input block, the 64 -> 1 deformable layer)."""
    if mode == "bf16" and tag.startswith("cl16_"):
        return "bf16_mfma", PEAK_BF16_MFMA_TFLOPS
    if mode == "bf16" and tag.startswith(("x3_", "deform64x3")):
        return "bf16_mfma/3", PEAK_SPLIT_BF16_TFLOPS
    return "fp32_mfma", PEAK_FP32_MFMA_TFLOPS


def sweep_leg(dbm, ctx, g, crops=5):
    """BASELINE config 5's unit of work, measured: one INTERIOR 288 x 288 crop of the continent sweep (deepbedmap.py:706-728;
    320 of the 396 crops are this size) -> 1144 x 1144, generator forward with resident inputs, fp32 and bf16, timed with HIP
    events on the library's stream over `crops` back-to-back crops after one warm-up.  FLOPs: SURVEY Appendix C's forward
    MACs per trunk pixel x the crop's 286 x 286 trunk pixels."""
    lib = dbm._lib.lib()
    h = w = 288
    r = np.random.RandomState(7)
    ins = [dbm.to_device(r.rand(1, c, m * h, m * w).astype(np.float32), ctx) for c, m in ((1, 1), (1, 10), (2, 2), (1, 1))]
    flop = 2.0 * G_FWD_MAC_PER_TILE / 81.0 * (h - 2) * (w - 2)
    in_bytes = 4.0 * h * w * (1 + 100 + 8 + 1)
    out_bytes = 4.0 * 16 * (h - 2) * (w - 2)
    res = {"crop": [h, w], "output": [4 * (h - 2), 4 * (w - 2)], "algorithmic_tflop_per_crop": flop / 1e12,
           "compulsory_bytes_per_crop": in_bytes + out_bytes, "crops_timed": crops,
           "crops_per_continent": 396, "resident_bytes_whole_continent": 4.0 * (2 * 4500 * 5500 + 45020 * 55020 + 2 * 9000 * 11000 + 18000 * 22000)}
    y = dbm.DeviceArray((1, 1, 4 * (h - 2), 4 * (w - 2)), ctx)  # (allocated once: dbm_malloc synchronises the device)

    def fwd(flags):
        dbm._lib.check(lib.dbm_gen_forward(g._h, 1, h, w, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, y.ptr,
                                           dbm._lib.DEVICE_PTRS | flags), ctx.handle)

    for name, flags in (("fp32", 0), ("bf16", dbm._lib.BF16)):
        fwd(flags)
        ctx.synchronize()
        dbm._lib.check(lib.dbm_timer(ctx.handle, 0, None), ctx.handle)
        for _ in range(crops):
            fwd(flags)
        dbm._lib.check(lib.dbm_timer(ctx.handle, 1, None), ctx.handle)
        ms = C.c_double(0.0)
        dbm._lib.check(lib.dbm_timer(ctx.handle, 2, C.byref(ms)), ctx.handle)
        per = ms.value / crops
        tf = flop / (per * 1e-3) / 1e12
        res[name] = {"ms_per_crop": per, "tflops": tf,
                     "frac_of_mfma_peak": tf / (PEAK_BF16_MFMA_TFLOPS if name == "bf16" else PEAK_FP32_MFMA_TFLOPS),
                     "mfma_peak_tflops": PEAK_BF16_MFMA_TFLOPS if name == "bf16" else PEAK_FP32_MFMA_TFLOPS,
                     "compulsory_gbs": (in_bytes + out_bytes) / (per * 1e-3) / 1e9,
                     "s_per_continent_one_gpu": 396 * per * 1e-3}
        # one more crop with every MFMA launch bracketed (device synchronised around each): per-shape standalone durations with
        # their algorithmic FLOP / bytes; launches that are not MFMA kernels (im2col of the 10x input, packing, the bilinear
        # sampler's transposes) are the difference between `bracketed_ms` and ms_per_crop
        dbm._lib.check(lib.dbm_profile_begin_serial(ctx.handle), ctx.handle)
        fwd(flags)
        recs = ctx.profile_records()
        shapes = {}
        for r_ in recs:
            row = shapes.setdefault((r_["tag"], r_["flops"], r_["bytes"]), {"shape": r_["tag"], "workgroups": r_["wgs"], "launches": 0, "ms": 0.0, "gflop_per_launch": r_["flops"] / 1e9,
                                                                           "algorithmic_bytes_per_launch": r_["bytes"]})
            row["launches"] += 1
            row["ms"] += r_["ms"]
        rows = []
        for row in sorted(shapes.values(), key=lambda q: -q["ms"]):
            row["avg_us"] = 1e3 * row["ms"] / row["launches"]
            row["tflops"] = row["gflop_per_launch"] * row["launches"] / row["ms"] if row["ms"] > 0 else 0.0
            row["roof"], row["roof_tflops"] = sweep_roof(name, row["shape"])
            row["frac_of_roof"] = row["tflops"] / row["roof_tflops"]
            row["algorithmic_gbps"] = row["algorithmic_bytes_per_launch"] * row["launches"] / (row["ms"] * 1e-3) / 1e9 if row["ms"] > 0 else 0.0
            rows.append({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()})
        res[name]["bracketed_ms"] = sum(r_["ms"] for r_ in recs)
        res[name]["per_shape_standalone"] = rows
    # equal-shape crops eight at a time (predict_tiled_resident(crops_per_batch=8)): the same work per crop in fuller launches
    nb = 8
    ins = [dbm.to_device(r.rand(nb, c, m * h, m * w).astype(np.float32), ctx) for c, m in ((1, 1), (1, 10), (2, 2), (1, 1))]
    y = dbm.DeviceArray((nb, 1, 4 * (h - 2), 4 * (w - 2)), ctx)

    def fwd8():
        dbm._lib.check(lib.dbm_gen_forward(g._h, nb, h, w, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, y.ptr,
                                           dbm._lib.DEVICE_PTRS | dbm._lib.BF16), ctx.handle)

    fwd8()
    ctx.synchronize()
    dbm._lib.check(lib.dbm_timer(ctx.handle, 0, None), ctx.handle)
    for _ in range(2):
        fwd8()
    dbm._lib.check(lib.dbm_timer(ctx.handle, 1, None), ctx.handle)
    ms = C.c_double(0.0)
    dbm._lib.check(lib.dbm_timer(ctx.handle, 2, C.byref(ms)), ctx.handle)
    per = ms.value / (2 * nb)
    res["bf16"]["batch8"] = {"ms_per_crop": per, "tflops": flop / (per * 1e-3) / 1e12,
                             "frac_of_mfma_peak": flop / (per * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                             "s_per_continent_one_gpu": 396 * per * 1e-3}
    return res


def resident_grid(dbm, ctx, r, c, h, w, lo, hi, band=100):
    """A (1, c, h, w) float32 grid in HBM without a host copy of it: `band` rows of U[lo, hi) are drawn on the host, uploaded once
    and replicated down the plane with device-to-device copies (the sweep's cost does not depend on the values; distinct rows
    within a band keep the crops from being constant)."""
    lib = dbm._lib.lib()
    band = min(band, h)
    src = dbm.to_device(r.uniform(lo, hi, (1, c, band, w)).astype(np.float32), ctx)
    grid = dbm.DeviceArray((1, c, h, w), ctx)
    for ch in range(c):
        for y in range(0, h, band):
            rows = min(band, h - y)
            dbm._lib.check(lib.dbm_memcpy2d_d2d(ctx.handle, C.c_void_p(grid.ptr + 4 * ((ch * h + y) * w)), 4 * w,
                                                C.c_void_p(src.ptr + 4 * (ch * band * w)), 4 * w, 4 * w, rows), ctx.handle)
    ctx.synchronize()
    return grid


def continent_leg(dbm, ctx, args, rank, world, comm):
    """BASELINE config 5: the whole-continent sweep of deepbedmap.py:689-741 -- 18000 x 22000 output pixels, 396 tiles of
    1000 x 1000 (320 interior 288 x 288 crops, 72 edge, 4 corner) dealt round-robin to the ranks (predict_tiled_resident(rank,
    world): every rank keeps the four grids resident, 10.7 GB, and writes its own tiles of its own canvas; no collective on the
    data path), generator forward in bf16, eight equal-shape crops per forward.  One warm-up sweep, then `--steps` timed sweeps
    (default 1 here) between barriers; the line's value is seconds per continent, max over ranks."""
    import torch

    S = dbm.Shape
    k = max(1, args.sweep_scale)
    H, W = 4500 // k, 5500 // k
    final = S(y=4 * H, x=4 * W)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dem_model import dem_generator

    g = dem_generator(dbm, seed=909)  # activations at the data's magnitude (metres), like a trained model's
    r = np.random.RandomState(1 + rank)
    grids = [resident_grid(dbm, ctx, r, 1, H, W, -2000, 2000), resident_grid(dbm, ctx, r, 1, 10 * H, 10 * W, -100, 4000, band=1000),
             resident_grid(dbm, ctx, r, 2, 2 * H, 2 * W, -10, 1000, band=200), resident_grid(dbm, ctx, r, 1, H, W, 0, 500)]
    n_tiles = len(dbm.tile_steps(final, S(y=1000, x=1000)))
    mine = sum(len(v) for v in dbm.group_tiles_by_crop_shape(final, S(y=1000, x=1000), S(y=1000, x=1000), S(y=18, x=18), rank, world).values())
    kw = dict(final_shape=final, dtype="bfloat16", download=False, crops_per_batch=8, rank=rank, world=world)
    for i in range(max(1, args.warmup)):  # warm-up sweeps (the first also does the one-time >= 0 clip, deepbedmap.py:663-665)
        canvas = dbm.predict_tiled_resident(g, *grids, clip=(i == 0), **kw)
        del canvas
    sweeps = max(1, args.steps)
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(sweeps):
        canvas = dbm.predict_tiled_resident(g, *grids, **kw)
    torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    dt = (time.perf_counter() - t0) / sweeps
    if comm is not None:
        dt = comm.max_over_ranks(dt)
    ok = True
    if k > 1 or rank == 0:   # the rank's own tiles are finite, nothing else was written (checked on a corner block of the canvas)
        blk = dbm.DeviceArray((1, min(final.y, 2152), min(final.x, 2152)), ctx)
        dbm._lib.check(dbm._lib.lib().dbm_memcpy2d_d2d(ctx.handle, C.c_void_p(blk.ptr), 4 * blk.shape[2], C.c_void_p(canvas.ptr), 4 * final.x,
                                                      4 * blk.shape[2], blk.shape[1]), ctx.handle)
        a = blk.get()[0]
        # (tile 0 -- output rows / columns 76 .. 1000 -- is rank 0's; the 76-pixel frame is never written)
        ok = bool(np.isnan(a[:76]).all() and np.isnan(a[:, :76]).all() and (np.isfinite(a[76:1000, 76:1000]).all() if rank == 0 else True))
    if rank != 0:
        return None
    flop = 2.0 * G_FWD_MAC_PER_TILE / 81.0 * sum((h - 2) * (w - 2) * len(t) for (h, w), t in
                                                  dbm.group_tiles_by_crop_shape(final, S(y=1000, x=1000), S(y=1000, x=1000), S(y=18, x=18)).items())
    return {"metric": "whole-continent tiled inference sweep, seconds (bf16 generator forward, tiles resident in HBM)", "value": _r(dt),
            "unit": "s", "n_gpus": world, "steps": sweeps, "warmup": max(1, args.warmup), "ms_per_step": _r(1e3 * dt), "higher_is_better": False, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"continent sweep {final.y} x {final.x} px, {n_tiles} tiles of 1000 x 1000, 12 RRDB, bf16 (BASELINE config 5)",
                       "crops_per_forward": 8, "tiles_rank0": mine, "resident_gb_per_rank": _r(sum(4.0 * a.size for a in grids) / 1e9 + 4.0 * final.y * final.x / 1e9, 2),
                       "parallelism": f"tiles round-robin over {world} ranks, no collective", "env": dbm_env()},
            "s_per_continent": _r(dt), "ranks": world, "tiles_per_s": _r(n_tiles / dt, 2), "algorithmic_tflop": _r(flop / 1e12, 1),
            "roofline": {"bound": "mfma", "achieved": _r(flop / dt / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": _r(flop / dt / 1e12 / (PEAK_BF16_MFMA_TFLOPS * world)), "traffic": None},
            "canvas_check_ok": ok}


def dbm_env(environ=None):
    """Every DBM_* variable of the environment (they select kernel variants / schedules; recorded in the line's config.env)."""
    environ = os.environ if environ is None else environ
    return {k: environ[k] for k in sorted(environ) if k.startswith("DBM_")}


def refuse_work_skipping_env(environ=None):
    """A measurement switch that skips work (libdbm_measure.so's ablations, or a library override) makes the timed region
    something other than the training iteration: bench.py does not run with one set."""
    bad = [k for k in dbm_env(environ) if k in WORK_SKIPPING_ENV]
    if bad:
        raise SystemExit(f"bench.py: refusing to run with work-skipping / library-override switches set: {bad}")


def _r(v, n=4):
    return round(v, n) if isinstance(v, float) else v


def compose_line(*, tiles, dt, steps, warmup, world, batch, ev_ms, config, fam, traffic=None, comm_stats=None, sweep=None, shared=None,
                 cpu=None, continent=None, tables_path=None, env=None, sync_metrics_ms=None):
    """The ONE JSON line rank 0 prints, as a dict: scalars only (the per-shape tables of the run go to bench_tables.json and to
    stderr).  `fam`: one dict per kernel family of the roofline leg (key, ms_per_step, launches_per_step, flop, bytes per step,
    standalone_ms_per_step).  json.dumps of the result stays below MAX_LINE_BYTES (fit_line enforces it)."""
    def tf(flop, ms):
        return flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

    dom = max(fam, key=lambda f: f["ms_per_step"])  # the dominant kernel = most summed launch time in one step
    n = max(dom["launches_per_step"], 1)
    ach, ach_s = tf(dom["flop"], dom["ms_per_step"]), tf(dom["flop"], dom["standalone_ms_per_step"])
    roof = {"bound": "mfma", "kernel": dom["key"], "achieved": _r(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": _r(ach / PEAK_FP32_MFMA_TFLOPS), "traffic": None,
            "frac_standalone": _r(ach_s / PEAK_FP32_MFMA_TFLOPS), "achieved_standalone": _r(ach_s, 3),
            "frac_step": _r(GFLOP_PER_TILE * batch * world / (dt / steps) / 1e3 / (PEAK_FP32_MFMA_TFLOPS * world)),
            "launches_per_step": dom["launches_per_step"], "avg_launch_us": _r(1e3 * dom["ms_per_step"] / n, 2),
            "standalone_avg_launch_us": _r(1e3 * dom["standalone_ms_per_step"] / n, 2),
            "algorithmic_gflop_per_launch": _r(dom["flop"] / n / 1e9), "algorithmic_bytes_per_launch": _r(dom["bytes"] / n, 0)}
    if traffic is not None:
        roof["traffic"] = traffic.get("hbm_bytes_per_launch")
        if roof["traffic"]:
            # ONE denominator: the ratio of the two figures THIS line states, traffic / algorithmic_bytes_per_launch
            # (the counter pass's own launch-weighted join stays in profiles/<round>/traffic_pmc.json)
            roof["traffic_over_algorithmic_bytes"] = _r(roof["traffic"] / max(dom["bytes"] / n, 1.0), 3)
        roof["traffic_source"] = traffic.get("source")
    roof["other_kernels"] = [{"kernel": f["key"], "ms": _r(f["ms_per_step"], 3), "ms_standalone": _r(f["standalone_ms_per_step"], 3),
                              "launches": f["launches_per_step"], "frac": _r(tf(f["flop"], f["ms_per_step"]) / PEAK_FP32_MFMA_TFLOPS),
                              "frac_standalone": _r(tf(f["flop"], f["standalone_ms_per_step"]) / PEAK_FP32_MFMA_TFLOPS)}
                             for f in fam if f is not dom and f["launches_per_step"] > 0]
    fwd = next((f for f in fam if f["key"] == "trunk_fused_kernel<helper>" and f["launches_per_step"] > 0), None) or \
        next((f for f in fam if f["key"] == "trunk_fused_kernel<retained>" and f["launches_per_step"] > 0), None)
    if fwd is not None:  # SURVEY 8d's second figure: the RRDB trunk forward of one 64-tile batch against the fp32 MFMA roof
        k = max(fwd["launches_per_step"], 1)
        roof["rrdb_forward"] = {"form": fwd["key"], "ms_per_batch": _r(fwd["standalone_ms_per_step"] / k),
                                "frac_standalone": _r(tf(fwd["flop"], fwd["standalone_ms_per_step"]) / PEAK_FP32_MFMA_TFLOPS)}
    out = {"metric": baseline_metric(), "value": _r(tiles / dt, 2), "unit": "tiles/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": _r(1e3 * dt / steps), "ms_per_step_hip_events": _r(ev_ms / steps), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": dict(config, env=dbm_env() if env is None else env), "roofline": roof}
    if comm_stats is not None:
        out["rccl_ranks"] = comm_stats.get("world")
        out["config"]["gradient_exchange"] = comm_stats
    if cpu is not None:
        c = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "host_cpus", "cpu_model", "wall_s", "g_forward_n1_ms", "g_forward_n64_ms")
             if k in cpu}
        c["value"] = _r(c["value"], 3)
        if "torch_cpu" in cpu:
            t = cpu["torch_cpu"]
            c["torch_cpu"] = {"value": _r(t["value"], 2), "threads": t.get("threads"), "sample": t.get("sample")} if "value" in t else t
        out["cpu_baseline"] = c
    extras = {}
    if sweep is not None:
        if "error" in sweep:
            extras["sweep"] = {"error": sweep["error"][:200]}
        else:
            sw = {"crop": sweep["crop"], "algorithmic_tflop_per_crop": _r(sweep["algorithmic_tflop_per_crop"])}
            for name in ("fp32", "bf16"):
                if name in sweep:
                    sw[name] = {k: _r(sweep[name][k]) for k in ("ms_per_crop", "tflops", "frac_of_mfma_peak", "s_per_continent_one_gpu")}
            if "batch8" in sweep.get("bf16", {}):
                sw["bf16"]["batch8_ms_per_crop"] = _r(sweep["bf16"]["batch8"]["ms_per_crop"])
            extras["sweep"] = sw
    if continent is not None:
        extras["continent"] = continent
    if shared is not None:
        extras["one_generator_forward"] = ({k: _r(shared[k]) for k in ("ms_per_step", "tiles_per_s")} if "error" not in shared
                                           else {"error": shared["error"][:200]})
    if sync_metrics_ms is not None:   # the reference's five float(...) per minibatch (srgan_train.py:1166, 1259-1263): one D2H read per step
        extras["sync_metrics_ms_per_step"] = _r(sync_metrics_ms)
    if extras:
        out["extras"] = extras
    if tables_path:
        out["tables"] = os.path.relpath(tables_path, ROOT) if os.path.abspath(tables_path).startswith(ROOT + os.sep) else tables_path
    return out


def fit_line(out, limit=MAX_LINE_BYTES):
    """json.dumps(out), guaranteed below `limit` bytes: optional detail is dropped (least important first) until it fits."""
    droppable = [("roofline", "traffic_source"), ("cpu_baseline", "sample"), ("cpu_baseline", "cpu_model"), ("extras", "one_generator_forward"),
                 ("roofline", "other_kernels"), ("config", "env"), ("extras", "sweep"), ("extras", "continent"), ("config", "gradient_exchange")]
    line = json.dumps(out, separators=(",", ":"))
    dropped = []
    for a, b in droppable:
        if len(line.encode()) < limit:
            break
        if isinstance(out.get(a), dict) and b in out[a]:
            if (a, b) == ("config", "env"):   # never silently: the NAMES of the switches stay
                out[a][b] = sorted(out[a][b])
            else:
                del out[a][b]
            dropped.append(f"{a}.{b}")
            out["dropped_for_length"] = dropped
            line = json.dumps(out, separators=(",", ":"))
    assert len(line.encode()) < limit, len(line)
    return line


def spawn_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher: this process has not touched the GPU (torch is not even imported
    yet), so it starts N fresh children -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what `python -m torch.distributed.run --nproc-per-node N` would set -- waits for them and
    forwards rank 0's JSON line.  Nothing is exec'ed or re-exec'ed.  Returns the exit code."""
    import socket
    import subprocess

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's pipe is drained by a thread WHILE the ranks run (a full pipe would block its write() and, behind it, every
    # rank's final barrier); a rank that dies leaves the others waiting in a collective or the rendezvous: end them (these
    # exact children only)
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.1)
    codes = [p.wait() for p in procs]
    reader.join(timeout=30)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def launcher_selftest(args):
    """What a rank does under --selftest-launcher: the rendezvous, barrier and max-over-ranks plumbing of the real run on
    the CPU (gloo), then ONE JSON line from rank 0."""
    import torch

    from deepbedmap_amd.parallel import DataParallel

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("DBM_SELFTEST_FAIL_RANK") == str(rank):
        return 3  # a rank that dies must turn into a non-zero exit code of the launcher
    comm = DataParallel(backend="gloo", device="cpu")
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = 1
    if world > 1:
        comm.dist.all_reduce(seen)
    slowest = comm.max_over_ranks(float(rank))
    comm.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_seen": seen.tolist(), "max_over_ranks": slowest,
                          "local_rank_env": os.environ.get("LOCAL_RANK"), "master": os.environ.get("MASTER_ADDR")}), flush=True)
    if world > 1:
        comm.dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200; 1 sweep with --sweep-continent)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 10; 1 sweep with --sweep-continent)")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="tiles per GPU (BASELINE: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the inference leg (extras.sweep: ms per 288x288 crop, fp32 and bf16)")
    ap.add_argument("--no-deterministic", action="store_true",
                    help="cudnn_deterministic=False: fp32 atomics instead of ordered gradient folds (the reference trains "
                         "with cudnn_deterministic=True, srgan_train.py:69, and so does the headline configuration)")
    ap.add_argument("--sync-batch-stats", action="store_true",
                    help="N > 1: BatchNorm / RaGAN statistics over the global batch (exactly one process at batch N*64; slower)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not enqueue the G-step's generator forward underneath the D-step's discriminator passes")
    ap.add_argument("--share-generator-forward", action="store_true",
                    help="NOT the headline configuration: reuse the D-step's generator forward in the G-step")
    ap.add_argument("--dist-backend", default=None, choices=["rccl", "nccl", "gloo"],
                    help="N > 1: rccl (default) = libdbm's native communicator, gradient buckets overlapped with the backward "
                         "passes; nccl = torch.distributed's RCCL, one all-reduce after each backward (round-1 form)")
    ap.add_argument("--force-comm", action="store_true",
                    help="single GPU only: attach a ONE-rank native RCCL communicator and treat it as active (DBM_COMM_FORCE_WORLD1=1): "
                         "the data-parallel schedule -- bucketed ncclAllReduce calls on the exchange stream, persistent launches at 192 "
                         "workgroups, optimizers behind the exchange events -- measured on one GPU")
    ap.add_argument("--no-fused-iteration", action="store_true",
                    help="the two step calls + two optimizer calls per minibatch instead of dbm_train_iteration (which, since "
                         "round 3, also serves data-parallel runs on the rccl / gloo-hook backends)")
    ap.add_argument("--sync-metrics", action="store_true",
                    help="fetch the five metrics to the host after every minibatch (the reference's float(...) pattern) instead "
                         "of once at the end of the run")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="no GPU work: the ranks only rendezvous over gloo on the CPU and rank 0 prints a JSON line "
                         "(tests/test_parallel_gloo.py drives the self-spawn path with it)")
    ap.add_argument("--sweep-continent", action="store_true",
                    help="BASELINE config 5 instead of the training iteration: the whole 18000 x 22000 sweep (396 tiles dealt round-robin "
                         "to the ranks, grids resident in HBM, bf16), one compact line with s_per_continent")
    ap.add_argument("--sweep-scale", type=int, default=1, help="--sweep-continent on a 1/k-scale area (tests)")
    ap.add_argument("--tables", default=TABLES_PATH,
                    help="where the per-shape tables of THIS run go (default bench_tables.json; profiling passes give their own path so "
                         "that they do not overwrite the tables of the unprofiled run)")
    ap.add_argument("--no-continent", action="store_true", help="skip extras.continent (one full 18000 x 22000 bf16 sweep on this GPU)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 1 if args.sweep_continent else 200
    if args.warmup is None:
        args.warmup = 1 if args.sweep_continent else 10
    refuse_work_skipping_env()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.selftest_launcher:
        sys.exit(launcher_selftest(args))

    # Contract: rank 0 prints ONE JSON line on stdout.  Libraries write banners to file descriptor 1 from C (RCCL prints its
    # version block when the first communicator is created): everything that is not the result goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    import deepbedmap_amd as dbm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if args.force_comm and world == 1:
        os.environ["DBM_COMM_FORCE_WORLD1"] = "1"   # (read by the library at its first exchange decision)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    comm = (dbm.DataParallel(backend=args.dist_backend or ("rccl" if args.force_comm else None), sync_batch_stats=args.sync_batch_stats)
            if (world > 1 or args.force_comm) else None)
    ctx = dbm.Context(local_rank)
    dbm._lib._default_ctx = ctx
    # multi-GPU: libdbm enqueues on the stream torch issues its RCCL collectives on (stream-ordered, no host waits);
    # single GPU: the context's own stream (torch.cuda.synchronize() below is device-wide)
    if comm is not None and not args.sweep_continent:   # (the sweep has no collective: the group only carries barriers)
        comm.attach(ctx)

    if args.sweep_continent:
        line = continent_leg(dbm, ctx, args, rank, world, comm)
        if rank == 0:
            os.write(result_fd, (fit_line(line) + "\n").encode())
        if comm is not None:
            comm.barrier()
        return

    np.random.seed(1234)  # identical initial weights on every rank (and broadcast below for good measure)
    g, g_opt, d, d_opt = dbm.compile_srgan_model(num_residual_blocks=N_RRDB, residual_scaling=0.1,
                                                 learning_rate=1.6e-4)
    if comm is not None:
        comm.broadcast_params(g)
        comm.broadcast_params(d)
    batch = dbm.device_batch(synthetic_batch(args.batch, 42 + rank), ctx)  # inputs resident in HBM before timing

    dbm.global_config.cudnn_deterministic = not args.no_deterministic
    prefetch = not (args.share_generator_forward or args.no_prefetch)

    # One step = one minibatch of deepbedmap_amd.trainer's training loop (srgan_train.py:1286-1309): D-step + G-step, the
    # five metrics written to a device-resident log that the host reads once at the end (trainer: once per epoch, where
    # their only consumer, the per-epoch mean, runs); --sync-metrics fetches them after every minibatch instead.
    log = None if args.sync_metrics else dbm.MetricsLog(ctx, rows=args.steps + args.warmup + 4)

    def step():
        return dbm.train_minibatch(batch, g, g_opt, d, d_opt, comm=comm, share_generator_forward=args.share_generator_forward,
                                   prefetch_generator_forward=prefetch, log=log, fused=not args.no_fused_iteration)

    lib = dbm._lib.lib()
    for _ in range(args.warmup):
        step()
    # (round 6: the last warm-up iteration's deferred eval-mode discriminator pass -- include/dbm.h, dbm_train_iteration -- is enqueued
    #  by the next library call: flush it HERE, outside the timed region.  The K timed iterations then contain exactly K such passes:
    #  K - 1 beside the following iteration's forwards, the last one alone, flushed by dbm_timer(1) / log.fetch() below.)
    ctx.synchronize()
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dbm._lib.check(lib.dbm_timer(ctx.handle, 0, None), ctx.handle)  # HIP events on the library's main stream as well
    for _ in range(args.steps):
        step()
    dbm._lib.check(lib.dbm_timer(ctx.handle, 1, None), ctx.handle)
    metrics_rows = log.fetch() if log is not None else None  # (inside the timed region: the epoch's single read-back)
    torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    dt = time.perf_counter() - t0
    ev_ms = C.c_double(0.0)
    dbm._lib.check(lib.dbm_timer(ctx.handle, 2, C.byref(ev_ms)), ctx.handle)
    if comm is not None:
        dt = comm.max_over_ranks(dt)
    comm_stats = None
    if comm is not None:
        cw, cb, cc = C.c_int(0), C.c_size_t(0), C.c_size_t(0)
        dbm._lib.check(lib.dbm_comm_stats(ctx.handle, C.byref(cw), C.byref(cb), C.byref(cc), 0), ctx.handle)
        comm_stats = {"backend": comm.backend, "world": cw.value, "bytes_per_step": cb.value / max(args.steps + args.warmup, 1),
                      "collectives_per_step": cc.value / max(args.steps + args.warmup, 1)}

    # ---- roofline leg (outside the timed region): hipEvent-bracketed launches of the dominant kernel ----
    assert metrics_rows is None or (len(metrics_rows) == args.steps + args.warmup and np.isfinite(metrics_rows[:, :5]).all())
    dbm._lib.check(lib.dbm_profile_begin(ctx.handle), ctx.handle)
    step()
    recs = ctx.profile_records()
    # the same brackets with the device synchronised around each launch: standalone durations
    dbm._lib.check(lib.dbm_profile_begin_serial(ctx.handle), ctx.handle)
    step()
    recs_s = ctx.profile_records()
    FAMILIES = [   # (key in the line, what it is) -- family index = KernelProfiler family (csrc/dbm_internal.h)
        ("per_layer_conv_family", "every convolution that is one launch per layer, forward + data gradient: igemm_conv_kernel / igemm_pm_kernel / conv_tile_kernel / "
                                  "input_block_fused_kernel / disc_deep_kernel + the fused deformable-convolution GEMMs (v_mfma_f32_32x32x2_f32)"),
        ("wgrad_kernel", "weight gradients: wgrad_wave_dma_kernel, wgrad_band_dma_kernel, wgrad_direct_kernel, wgrad_kernel"),
        ("trunk_fused_kernel<retained>", "RRDB trunk forward of a retained pass, one persistent launch"),
        ("trunk_fused_bwd_kernel", "RRDB trunk data-gradient chain, persistent"),
        ("trunk_fused_kernel<helper>", "RRDB trunk forward of a pass that keeps nothing (the D-step's fakes, inference): a helper workgroup per image"),
    ]
    fam = []
    for i, (key, label) in enumerate(FAMILIES):
        mine, mine_s = [r for r in recs if r["family"] == i], [r for r in recs_s if r["family"] == i]
        fam.append({"key": key, "label": label, "ms_per_step": sum(r["ms"] for r in mine), "launches_per_step": len(mine),
                    "flop": sum(r["flops"] for r in mine), "bytes": sum(r["bytes"] for r in mine),
                    "standalone_ms_per_step": sum(r["ms"] for r in mine_s)})
    per_shape = shape_table(recs, recs_s, [k for k, _ in FAMILIES])

    # ---- inference leg (outside the timed region, rank 0 of a single-GPU run): BASELINE config 5's unit of work ----
    sweep = None
    if rank == 0 and world == 1 and not args.no_sweep:
        try:
            sweep = sweep_leg(dbm, ctx, g)
        except Exception as e:  # pragma: no cover
            sweep = {"error": repr(e)}

    # ---- the same iteration with ONE generator forward (outside the timed region, single GPU; never the headline) ----
    shared = None
    if rank == 0 and world == 1 and not args.share_generator_forward and not args.no_sweep:
        try:
            log_s = dbm.MetricsLog(ctx, rows=64)

            def step_shared():
                return dbm.train_minibatch(batch, g, g_opt, d, d_opt, share_generator_forward=True, prefetch_generator_forward=False,
                                           log=log_s, fused=not args.no_fused_iteration)
            for _ in range(5):
                step_shared()
            ctx.synchronize()   # (flushes the deferred eval-mode pass: the loop below then holds exactly n_shared of them)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            n_shared = 40
            for _ in range(n_shared):
                step_shared()
            ctx.synchronize()
            torch.cuda.synchronize()
            dts = (time.perf_counter() - ts) / n_shared
            shared = {"ms_per_step": 1e3 * dts, "tiles_per_s": args.batch / dts, "steps": n_shared,
                      "note": "opt-in share_generator_forward=True: the G-step reuses the D-step's generator forward (same weights, "
                              "same inputs: the two-forward results up to fp32 rounding, tests/test_gpu_model.py::"
                              "test_shared_generator_forward_is_equivalent); 3 G_f + 7 D_f = 450 GFLOP per iteration instead of 539.5. "
                              "Not the reference's call sequence (srgan_train.py:1131 and :1222 both run the generator), so `value` is "
                              "the two-forward iteration."}
        except Exception as e:  # pragma: no cover
            shared = {"error": repr(e)}

    # ---- the same iteration with the reference's per-minibatch read-back (outside the timed region, single GPU) ----
    sync_ms = None
    if rank == 0 and world == 1 and not args.sync_metrics and not args.no_sweep:
        try:
            def step_sync():
                return dbm.train_minibatch(batch, g, g_opt, d, d_opt, share_generator_forward=args.share_generator_forward,
                                           prefetch_generator_forward=prefetch, log=None, fused=not args.no_fused_iteration)
            for _ in range(3):
                step_sync()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            n_sync = max(1, min(args.steps, 50))
            for _ in range(n_sync):
                step_sync()
            torch.cuda.synchronize()
            sync_ms = 1e3 * (time.perf_counter() - ts) / n_sync
        except Exception as e:  # pragma: no cover
            print("sync-metrics leg failed: " + repr(e), file=sys.stderr)

    # ---- BASELINE config 5 as far as one GPU reaches (outside the timed region): ONE full 18000 x 22000 sweep, bf16, resident ----
    continent = None
    if rank == 0 and world == 1 and not args.no_sweep and not args.no_continent:
        try:
            cl = continent_leg(dbm, ctx, argparse.Namespace(sweep_scale=1, warmup=1, steps=1), 0, 1, None)
            continent = {k: cl[k] for k in ("s_per_continent", "tiles_per_s", "canvas_check_ok")}
            continent["frac_of_bf16_mfma_peak"] = cl["roofline"]["frac"]
            continent["area_px"] = [18000, 22000]
        except Exception as e:  # pragma: no cover
            continent = {"error": repr(e)[:200]}

    if rank == 0:
        config = {"workload": "full ESRGAN training iteration (D-step + G-step, fwd+bwd+Adam), 12 RRDB, 11x11 -> 36x36 tiles, fp32",
                  "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                  "generator_forwards_per_iteration": 1 if args.share_generator_forward else 2,
                  "g_step_forward_prefetched_under_d_step": bool(prefetch),
                  "cudnn_deterministic": bool(dbm.global_config.cudnn_deterministic),
                  "fused_iteration_call": bool(not args.no_fused_iteration and prefetch and
                                               (comm is None or (comm.exchanges_in_step(ctx) and not args.sync_batch_stats))),
                  "metrics_read_back": "every minibatch" if args.sync_metrics else "once per run",
                  "sync_batch_stats": bool(args.sync_batch_stats and world > 1)}
        dom_key = max(fam, key=lambda f: f["ms_per_step"])["key"]
        traffic = None
        try:  # HBM bytes per launch of the dominant kernel: PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE, from the committed pass
            with open(TRAFFIC_JSON) as f:
                tjs = json.load(f)
                tj = tjs.get(dom_key.split("<")[0]) or tjs[{"per_layer_conv_family": "igemm_conv_kernel"}[dom_key]]   # (key of the files up to round 5)
                traffic = {"hbm_bytes_per_launch": tj["hbm_bytes_per_launch"], "traffic_over_algorithmic": tj.get("traffic_over_algorithmic"),
                           "source": "static: " + os.path.relpath(TRAFFIC_JSON, ROOT) + " (rocprofv3 --pmc, separate passes; not this run)"}
        except Exception:
            pass
        cpu = cpu_baseline() if (not args.no_cpu_baseline and world == 1) else None
        tables = {"families": fam, "per_shape": per_shape, "sweep": sweep, "one_generator_forward": shared, "cpu_baseline": cpu,
                  "continent": continent, "sync_metrics_ms_per_step": sync_ms,
                  "note": "per_shape: every distinct launch shape of the step (tag = layer geometry), sorted by standalone time"}
        tables_path = None
        try:
            with open(args.tables, "w") as f:
                json.dump(tables, f, indent=1)
            tables_path = args.tables
        except OSError:
            pass
        if tables_path is None:   # (no writable directory: the tables go to stderr instead)
            print("bench tables: " + json.dumps(tables), file=sys.stderr, flush=True)
        out = compose_line(tiles=args.batch * world * args.steps, dt=dt, steps=args.steps, warmup=args.warmup, world=world, batch=args.batch,
                           ev_ms=ev_ms.value, config=config, fam=fam, traffic=traffic, comm_stats=comm_stats, sweep=sweep, shared=shared,
                           cpu=cpu, tables_path=tables_path, continent=continent, sync_metrics_ms=sync_ms)
        os.write(result_fd, (fit_line(out) + "\n").encode())
    if comm is not None:
        comm.barrier()


if __name__ == "__main__":
    main()
