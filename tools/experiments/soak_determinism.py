#!/usr/bin/env python3
"""Soak: the fused training iteration N times from one seed, twice -- parameters, Adam state and the metrics log must come out bitwise
equal (a race between the iteration's four streams would show up as a difference), the time-out counters must stay zero.
usage (GPU box): python tools/experiments/soak_determinism.py [iterations]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import deepbedmap_amd as dbm
from bench import synthetic_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500


def run():
    ctx = dbm.Context(0)
    dbm._lib._default_ctx = ctx
    np.random.seed(99)
    g, go, d, do = dbm.compile_srgan_model(12, 0.1, 1.6e-4)
    batches = [dbm.device_batch(synthetic_batch(64, 40 + k), ctx) for k in range(3)]
    log = dbm.MetricsLog(ctx, rows=n + 8)
    for i in range(n):
        dbm.train_minibatch(batches[i % 3], g, go, d, do, log=log)
    ctx.synchronize()
    h = hashlib.sha256()
    for m in (g, d):
        for k, v in sorted(m.serialize_dict().items()):
            h.update(np.ascontiguousarray(v).tobytes())
    rows = log.fetch()
    h.update(np.ascontiguousarray(rows).tobytes())
    return h.hexdigest(), ctx.timeout_info(), bool(np.isfinite(rows[:, :5]).all()), rows[-1, :5].tolist()


a = run()
b = run()
print("run 1", a)
print("run 2", b)
print("BITWISE EQUAL" if a[0] == b[0] else "DIFFERENT")
sys.exit(0 if a[0] == b[0] and a[2] and b[2] else 1)
