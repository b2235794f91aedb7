#!/usr/bin/env python3
"""N fused training iterations at the benchmark's configuration and nothing else (for rocprofv3 traces of the step)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm  # noqa: E402
from bench import synthetic_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = dbm.Context(0)
dbm._lib._default_ctx = ctx
np.random.seed(1234)
g, go, d, do = dbm.compile_srgan_model(12, 0.1, 1.6e-4)
batch = dbm.device_batch(synthetic_batch(64, 42), ctx)
log = dbm.MetricsLog(ctx, rows=n + 8)
for _ in range(5):
    dbm.train_minibatch(batch, g, go, d, do, log=log)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    dbm.train_minibatch(batch, g, go, d, do, log=log)
ctx.synchronize()
print("ms_per_step %.4f" % ((time.perf_counter() - t0) / n * 1e3), "timeout_info", ctx.timeout_info())
