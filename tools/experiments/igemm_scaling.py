#!/usr/bin/env python3
"""Fixed cost and K-loop rate of the per-layer convolution kernel: 3x3 stride-1 layers at batch 64 through dbm_op_conv2d with the
serialised profiler brackets (standalone durations), Cin swept at fixed grids."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm
from deepbedmap_amd import _lib
ctx = dbm.Context(0); _lib._default_ctx = ctx
lib = _lib.lib()
rs = np.random.RandomState(0)
N = 64
for (H, W, O) in ((36, 36, 64), (36, 36, 32), (18, 18, 128), (18, 18, 64), (9, 9, 128), (4, 4, 256)):
    for Cin in (32, 64, 128, 256, 512):
        x = dbm.to_device(rs.rand(N, Cin, H, W).astype(np.float32), ctx)
        w = dbm.to_device((rs.rand(O, Cin, 3, 3) - 0.5).astype(np.float32), ctx)
        b = dbm.to_device(rs.rand(O).astype(np.float32), ctx)
        y = dbm.DeviceArray((N, O, H, W), ctx)
        def run():
            _lib.check(lib.dbm_op_conv2d(ctx.handle, x.ptr, w.ptr, b.ptr, y.ptr, N, Cin, H, W, O, 3, 1, 1, 0, 1), ctx.handle)
        run(); run()
        best = 1e9
        for _ in range(3):
            _lib.check(lib.dbm_profile_begin_serial(ctx.handle), ctx.handle)
            run()
            recs = ctx.profile_records()
            best = min(best, sum(r["ms"] for r in recs if r["tag"].startswith("c")))
        gf = 2.0 * N * H * W * O * Cin * 9 / 1e9
        print("%2dx%-2d Cout %3d Cin %3d: %7.1f us  %6.1f TFLOP/s (%.2f of peak)  %s" % (H, W, O, Cin, best * 1e3, gf / best, gf / best / 157.3, recs[-1]["tag"]), flush=True)
