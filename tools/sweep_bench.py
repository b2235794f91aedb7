"""Timing of the tiled area inference on a 1/6-scale area (12 RRDB): per-tile upload vs HBM-resident, fp32 vs bf16.

    python tools/sweep_bench.py
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import deepbedmap_amd as dbm
ctx = dbm.Context(0); dbm._lib._default_ctx = ctx
np.random.seed(0)
g = dbm.GeneratorModel(num_residual_blocks=12)
S = dbm.Shape
H, W = 750, 1000   # low-resolution pixels (full continent: 4500 x 5500)
r = np.random.RandomState(1)
X = r.rand(1, 1, H, W).astype(np.float32); W1 = r.rand(1, 1, 10*H, 10*W).astype(np.float32)
W2 = r.rand(1, 2, 2*H, 2*W).astype(np.float32); W3 = r.rand(1, 1, H, W).astype(np.float32)
final = S(y=4*H, x=4*W)
ref = None
for name, fn, dt in (("per-tile upload, fp32", dbm.predict_tiled, "float32"), ("HBM-resident, fp32", dbm.predict_tiled_resident, "float32"),
                     ("HBM-resident, bf16", dbm.predict_tiled_resident, "bfloat16")):
    fn(g, X, W1, W2, W3, final_shape=S(y=1000, x=1000), ary_shape=S(y=1000, x=1000), dtype=dt)  # warm-up on one tile
    ctx.synchronize(); t0 = time.perf_counter()
    Y = fn(g, X, W1, W2, W3, final_shape=final, dtype=dt)
    ctx.synchronize(); dt_s = time.perf_counter() - t0
    n = len(dbm.tile_steps(final, S(y=1000, x=1000)))
    inner = Y[:, 76:-76, 76:-76]
    if ref is None: ref = inner
    print(f"{name}: {n} tiles, {dt_s:.2f} s, {dt_s/n*1e3:.1f} ms per tile, finite {np.isfinite(inner).all()}, "
          f"max|diff to fp32|/max = {np.abs(inner - ref).max() / np.abs(ref).max():.2e}")
