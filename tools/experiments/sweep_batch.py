import os, sys, json, ctypes as C
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/deepbedmap_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import deepbedmap_amd as dbm
ctx = dbm.Context(0); dbm._lib._default_ctx = ctx
np.random.seed(1)
g = dbm.GeneratorModel(num_residual_blocks=12)
lib = dbm._lib.lib()
h = w = 288
for nb in (1, 2, 3, 4, 6, 8, 12, 16):
    r = np.random.RandomState(7)
    ins = [dbm.to_device(r.rand(nb, c, m * h, m * w).astype(np.float32), ctx) for c, m in ((1, 1), (1, 10), (2, 2), (1, 1))]
    y = dbm.DeviceArray((nb, 1, 4 * (h - 2), 4 * (w - 2)), ctx)
    for name, flags in (("fp32", 0), ("bf16", dbm._lib.BF16)):
        if name == "fp32" and nb > 2: continue
        def fwd():
            dbm._lib.check(lib.dbm_gen_forward(g._h, nb, h, w, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, y.ptr, dbm._lib.DEVICE_PTRS | flags), ctx.handle)
        fwd(); ctx.synchronize()
        dbm._lib.check(lib.dbm_timer(ctx.handle, 0, None), ctx.handle)
        for _ in range(3): fwd()
        dbm._lib.check(lib.dbm_timer(ctx.handle, 1, None), ctx.handle)
        ms = C.c_double(0.0)
        dbm._lib.check(lib.dbm_timer(ctx.handle, 2, C.byref(ms)), ctx.handle)
        print(nb, name, "ms per crop %.3f" % (ms.value / 3 / nb), flush=True)
    del ins, y
