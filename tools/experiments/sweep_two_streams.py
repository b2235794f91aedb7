#!/usr/bin/env python3
"""How much do two INDEPENDENT bf16 sweep forwards overlap on one GPU?  Two contexts (own streams, own copies of the generator), crops
enqueued alternately; per-crop time against one context alone.  (Feasibility probe for pipelining the tail of crop i under the trunk of
crop i + 1.)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import deepbedmap_amd as dbm

h = w = 288
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = dbm._lib.lib()
ctxs = [dbm.Context(0), dbm.Context(0)]
sets = []
for ctx in ctxs:
    dbm._lib._default_ctx = ctx
    np.random.seed(3)
    g = dbm.GeneratorModel(num_residual_blocks=12, residual_scaling=0.1, ctx=ctx)
    r = np.random.RandomState(7)
    ins = [dbm.to_device(r.rand(nb, c, m * h, m * w).astype(np.float32), ctx) for c, m in ((1, 1), (1, 10), (2, 2), (1, 1))]
    y = dbm.DeviceArray((nb, 1, 4 * (h - 2), 4 * (w - 2)), ctx)
    sets.append((ctx, g, ins, y))


def fwd(k):
    ctx, g, ins, y = sets[k]
    dbm._lib.check(lib.dbm_gen_forward(g._h, nb, h, w, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, y.ptr,
                                       dbm._lib.DEVICE_PTRS | dbm._lib.BF16), ctx.handle)


for k in (0, 1):
    fwd(k); sets[k][0].synchronize()
n = 8
t0 = time.perf_counter()
for _ in range(n):
    fwd(0)
sets[0][0].synchronize()
one = (time.perf_counter() - t0) / n / nb * 1e3
t0 = time.perf_counter()
for _ in range(n):
    fwd(0); fwd(1)
sets[0][0].synchronize(); sets[1][0].synchronize()
two = (time.perf_counter() - t0) / (2 * n) / nb * 1e3
print("crops per forward %d: one stream %.3f ms per crop, two streams %.3f ms per crop (%.1f %%)" % (nb, one, two, 100 * (two / one - 1)))
