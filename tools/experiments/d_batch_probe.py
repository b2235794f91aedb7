#!/usr/bin/env python3
"""Would one N = 128 launch per discriminator layer (real + fake batch together) beat two N = 64 launches?  Standalone durations of the
discriminator's convolution shapes (forward) at N = 64 and N = 128 through dbm_op_conv2d (serialised profiler brackets)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm
from deepbedmap_amd import _lib
ctx = dbm.Context(0); _lib._default_ctx = ctx
lib = _lib.lib()
rs = np.random.RandomState(0)
SHAPES = [(64, 36, 64, 4, 2), (64, 18, 128, 3, 1), (128, 18, 128, 4, 2), (128, 9, 128, 3, 1), (128, 9, 256, 4, 2), (256, 4, 256, 3, 1),
          (256, 4, 512, 4, 2), (512, 2, 512, 3, 1), (512, 2, 512, 4, 2)]
tot = {64: 0.0, 128: 0.0}
for (C, H, O, k, s) in SHAPES:
    row = []
    for N in (64, 128):
        x = dbm.to_device(rs.rand(N, C, H, H).astype(np.float32), ctx)
        w = dbm.to_device((rs.rand(O, C, k, k) - 0.5).astype(np.float32), ctx)
        b = dbm.to_device(rs.rand(O).astype(np.float32), ctx)
        OH = (H + 2 - k) // s + 1
        y = dbm.DeviceArray((N, O, OH, OH), ctx)
        def run():
            _lib.check(lib.dbm_op_conv2d(ctx.handle, x.ptr, w.ptr, b.ptr, y.ptr, N, C, H, H, O, k, s, 1, 0, 1), ctx.handle)
        run(); run()
        best = 1e9
        for _ in range(4):
            _lib.check(lib.dbm_profile_begin_serial(ctx.handle), ctx.handle)
            run()
            recs = ctx.profile_records()
            best = min(best, sum(r["ms"] for r in recs if r["family"] == 0))
        row.append(best * 1e3)
        tot[N] += best * 1e3
    print("c%d>%d k%d s%d %dx%d: N=64 %6.1f us   N=128 %6.1f us   ratio %.2f" % (C, O, k, s, H, H, row[0], row[1], row[1] / row[0]), flush=True)
print("sum: 2 x N=64 %.1f us, 1 x N=128 %.1f us" % (2 * tot[64], tot[128]))
