"""Where a training step goes: hipEvents at the phase boundaries of the main stream (dbm_phase_marks), unprofiled.

    python tools/phases.py [prefetch|narrow|share|atomics|fused]   (atomics: prefetch with cudnn_deterministic=False;
                                                                    fused: dbm_train_iteration, marks of several streams)
"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import deepbedmap_amd as dbm
from bench import synthetic_batch
ctx = dbm.Context(0); dbm._lib._default_ctx = ctx
np.random.seed(1)
g, go, d, do = dbm.compile_srgan_model(12, 0.1, 1.6e-4)
batch = dbm.device_batch(synthetic_batch(64, 42), ctx)
share = len(sys.argv) > 1 and sys.argv[1] == "share"
prefetch = len(sys.argv) > 1 and sys.argv[1] in ("prefetch", "narrow", "atomics")
dbm.global_config.cudnn_deterministic = not (len(sys.argv) > 1 and sys.argv[1] == "atomics")
class FakeComm:
    def allreduce_grads(self, model):
        return 1.0
comm = FakeComm() if len(sys.argv) > 1 and sys.argv[1] == "narrow" else None
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
def step():
    if fused:
        dbm.train_minibatch(batch, g, go, d, do, fused=True); return
    dbm.train_eval_discriminator(batch, g, d, do, share_generator_forward=share, prefetch_generator_forward=prefetch, comm=comm); dbm.train_eval_generator(batch, g, d, go, share_generator_forward=share)
for _ in range(4): step()
lib = dbm._lib.lib()
buf = C.create_string_buffer(1 << 16)
acc = {}
order = []
NREP = 5
for rep in range(NREP):
    ctx.synchronize()
    dbm._lib.check(lib.dbm_phase_marks(ctx.handle, 1, None, 0), ctx.handle)
    step()
    dbm._lib.check(lib.dbm_phase_marks(ctx.handle, 0, buf, len(buf)), ctx.handle)
    prev = 0.0
    seen = {}
    for line in buf.value.decode().strip().splitlines():
        name, ms = line.rsplit(" ", 1); ms = float(ms)
        k = seen.get(name, 0); seen[name] = k + 1
        key = f"{name}#{k}" if k else name
        if key not in acc: acc[key] = []; order.append(key)
        acc[key].append((ms - prev, ms)); prev = ms
for k in order:
    v = acc[k]
    print("%-40s +%7.3f ms   (at %7.3f)" % (k, sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)))
