#!/usr/bin/env python3
"""The LDS-tiled convolution (conv_tile.hip) against igemm_conv_kernel on single launches, for rocprofv3 (kernel trace / --pmc):
every shape is launched `reps` times through dbm_op_conv2d; DBM_CONV_TILE=0/1 is read per process.

    python tools/experiments/conv_tile_bench.py [reps]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm  # noqa: E402
from deepbedmap_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = dbm.Context(0)
dbm._lib._default_ctx = ctx
rs = np.random.RandomState(0)
SHAPES = [  # N, C, H, W, O, ups
    (64, 64, 36, 36, 64, 0), (64, 64, 36, 36, 18, 0), (64, 32, 36, 36, 64, 0), (64, 64, 18, 18, 64, 1),
    (64, 64, 18, 18, 128, 0), (64, 128, 18, 18, 64, 0), (64, 64, 18, 18, 64, 0), (64, 64, 9, 9, 64, 1),
]
for (N, C, H, W, O, ups) in SHAPES:
    x = dbm.to_device(rs.rand(N, C, H, W).astype(np.float32), ctx)
    w = dbm.to_device((rs.rand(O, C, 3, 3) - 0.5).astype(np.float32), ctx)
    b = dbm.to_device(rs.rand(O).astype(np.float32), ctx)
    y = dbm.DeviceArray((N, O, H << ups, W << ups), ctx)
    for _ in range(reps):
        _lib.check(_lib.lib().dbm_op_conv2d(ctx.handle, x.ptr, w.ptr, b.ptr, y.ptr, N, C, H, W, O, 3, 1, 1, ups, 1), ctx.handle)
ctx.synchronize()
print("done")
