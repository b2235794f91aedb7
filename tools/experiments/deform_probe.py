#!/usr/bin/env python3
"""Standalone durations of the fused deformable forward (64 -> 64, 36 x 36, batch 64) through dbm_op_deform_conv2d; with
libdbm_measure.so, DBM_DEFORM_ABL=1 makes every gather hit one cache-resident pixel (what the gathers' locality costs)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm
from deepbedmap_amd import _lib
ctx = dbm.Context(0); _lib._default_ctx = ctx
lib = _lib.lib()
rs = np.random.RandomState(0)
N, C, H, W, O = 64, 64, 36, 36, 64
for scale in (0.1, 1.0, 3.0):
    x = dbm.to_device(rs.rand(N, C, H, W).astype(np.float32), ctx)
    off = dbm.to_device(rs.normal(scale=scale, size=(N, 18, H, W)).astype(np.float32), ctx)
    w = dbm.to_device((rs.rand(O, C, 3, 3) - 0.5).astype(np.float32), ctx)
    b = dbm.to_device(rs.rand(O).astype(np.float32), ctx)
    y = dbm.DeviceArray((N, O, H, W), ctx)
    def run():
        _lib.check(lib.dbm_op_deform_conv2d(ctx.handle, x.ptr, off.ptr, w.ptr, b.ptr, y.ptr, N, C, H, W, O), ctx.handle)
    run(); run()
    best = 1e9
    for _ in range(4):
        _lib.check(lib.dbm_profile_begin_serial(ctx.handle), ctx.handle)
        run()
        recs = ctx.profile_records()
        best = min(best, sum(r["ms"] for r in recs if r["tag"].startswith("deform")))
    print("ABL=%s offsets sigma %.1f px: %6.1f us" % (os.environ.get("DBM_DEFORM_ABL", "0"), scale, best * 1e3), flush=True)
