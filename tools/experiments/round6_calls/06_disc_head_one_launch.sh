#!/bin/bash
# round 6, call 6: the discriminator's head (linear_1 -> LeakyReLU -> linear_2) as one launch per pass, forward and backward:
# parity (discriminator suites, bitwise against the two-launch form through the fused iteration's parameters), A/B
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c6; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_round5.py -x -q -m gpu -k "discriminator or iteration or soak or schedule" > $O/tests.log 2>&1; tail -3 $O/tests.log
M=$PWD/deepbedmap_amd/libdbm_measure.so
# bitwise: three fused iterations with both forms -> SHA of every parameter of both models
for e in "DBM_X=1" "DBM_DISC_HEAD_FUSED=0"; do env DBM_LIB=$M $e timeout 300 python3 - <<'PY'
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.getcwd())
import deepbedmap_amd as d
from bench import synthetic_batch
ctx = d.Context(0); d._lib._default_ctx = ctx
np.random.seed(5)
g, go, dm, do = d.compile_srgan_model(2, 0.3, 1e-3)
b = d.device_batch(synthetic_batch(64, 7), ctx)
m = [d.train_minibatch(b, g, go, dm, do, fused=True) for _ in range(3)]
h = hashlib.sha256()
for mod in (g, dm):
    for k, v in sorted(mod.serialize_dict().items()): h.update(np.ascontiguousarray(v).tobytes())
print(os.environ.get("DBM_DISC_HEAD_FUSED", "fused"), h.hexdigest()[:16], m[-1])
PY
done > $O/bitwise.txt 2>&1; cat $O/bitwise.txt
for rep in 1 2 3; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_DISC_HEAD_FUSED=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_head.txt 2>&1; cat $O/ab_head.txt
