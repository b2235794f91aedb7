#!/usr/bin/env python3
"""How long does the HOST need to enqueue one fused training iteration (the call returns when everything is queued)?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm
from bench import synthetic_batch
ctx = dbm.Context(0); dbm._lib._default_ctx = ctx
np.random.seed(1234)
g, go, d, do = dbm.compile_srgan_model(12, 0.1, 1.6e-4)
batch = dbm.device_batch(synthetic_batch(64, 42), ctx)
log = dbm.MetricsLog(ctx, rows=400)
for _ in range(5):
    dbm.train_minibatch(batch, g, go, d, do, log=log)
ctx.synchronize()
for n in (4, 16, 64):
    t0 = time.perf_counter()
    for _ in range(n):
        dbm.train_minibatch(batch, g, go, d, do, log=log)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print("n=%3d: host enqueue %.3f ms per iteration, until the device is done %.3f ms per iteration" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
