#!/usr/bin/env python3
"""conv_cl16_kernel: where do the ~10 us that do not scale with the input channels go?  DBM_CL16_ABL (read once per process):
1 = no MFMA loop, 2 = no epilogue, 4 = only chunk 0 staged.  One 286 x 286 plane, Cin = 64 / 160, Cout 32, standalone durations."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm
from deepbedmap_amd import _lib
ctx = dbm.Context(0); _lib._default_ctx = ctx
lib = _lib.lib()
rs = np.random.RandomState(0)
H = W = 286
for Cin, O in ((64, 32), (160, 32), (192, 64)):
    x = dbm.to_device(rs.rand(1, Cin, H, W).astype(np.float32), ctx)
    w = dbm.to_device((rs.rand(O, Cin, 3, 3) - 0.5).astype(np.float32), ctx)
    b = dbm.to_device(rs.rand(O).astype(np.float32), ctx)
    y = dbm.DeviceArray((1, O, H, W), ctx)
    def run():
        _lib.check(lib.dbm_op_conv2d_cl16(ctx.handle, x.ptr, w.ptr, b.ptr, None, 1.0, y.ptr, 1, Cin, H, W, O, 1), ctx.handle)
    run(); run()
    best = 1e9
    for _ in range(5):
        _lib.check(lib.dbm_profile_begin_serial(ctx.handle), ctx.handle)
        run()
        recs = ctx.profile_records()
        best = min(best, sum(r["ms"] for r in recs if r["tag"].startswith("cl16")))
    print("ABL=%s Cin %3d Cout %2d: %6.1f us" % (os.environ.get("DBM_CL16_ABL", "0"), Cin, O, best * 1e3), flush=True)
