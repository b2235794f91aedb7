# timing of the tiled area inference: per-tile upload vs HBM-resident, on a 1/6-scale area (12 RRDB, fp32)
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
for name, fn in (("per-tile upload", dbm.predict_tiled), ("HBM-resident", dbm.predict_tiled_resident)):
    fn(g, X, W1, W2, W3, final_shape=S(y=1000, x=1000), ary_shape=S(y=1000, x=1000))  # warm-up on one tile
    ctx.synchronize(); t0 = time.perf_counter()
    Y = fn(g, X, W1, W2, W3, final_shape=final)
    ctx.synchronize(); dt = time.perf_counter() - t0
    n = len(dbm.tile_steps(final, S(y=1000, x=1000)))
    print(f"{name}: {n} tiles, {dt:.2f} s, {dt/n*1e3:.1f} ms per tile, finite interior {np.isfinite(Y[:, 76:-76, 76:-76]).all()}")
