#!/usr/bin/env python3
"""Per-shape standalone / in-step times of one training iteration (the bracketed MFMA launches), filtered by a substring of the
kernel family or shape tag.  Works with libdbm_measure.so (DBM_LIB=..., ablation switches), which bench.py refuses.

    python tools/experiments/step_shapes.py [filter] [iterations]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deepbedmap_amd as dbm  # noqa: E402

flt = sys.argv[1] if len(sys.argv) > 1 else ""
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = dbm.Context(0)
dbm._lib._default_ctx = ctx
lib = dbm._lib.lib()
np.random.seed(1234)
g, go, d, do = dbm.compile_srgan_model(12, 0.1, 1.6e-4)
batch = dbm.device_batch(bench.synthetic_batch(64, 42), ctx)
log = dbm.MetricsLog(ctx, rows=n + 16)
step = lambda: dbm.train_minibatch(batch, g, go, d, do, log=log)  # noqa: E731
for _ in range(5):
    step()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
ctx.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
dbm._lib.check(lib.dbm_profile_begin(ctx.handle), ctx.handle)
step()
recs = ctx.profile_records()
dbm._lib.check(lib.dbm_profile_begin_serial(ctx.handle), ctx.handle)
step()
recs_s = ctx.profile_records()
rows = bench.shape_table(recs, recs_s, ["igemm", "wgrad", "trunk_fwd_retained", "trunk_bwd", "trunk_fwd_helper"])
env = {k: v for k, v in os.environ.items() if k.startswith("DBM_")}
print(f"env {env} step {ms:.3f} ms; sum standalone {sum(r['ms_standalone'] for r in rows):.3f} ms")
for r in rows:
    if flt in r["kernel"] or flt in r["shape"]:
        print(f"  {r['kernel']:20s} {r['shape']:32s} x{r['launches']:<2d} standalone {r['avg_us_standalone']:8.1f} us  in-step {1e3 * r['ms'] / r['launches']:8.1f} us  "
              f"frac {r['frac_mfma_standalone']:.3f}")
