#!/bin/bash
# round 6, call 19: the discriminator's deep end (conv_layer5..9 + BatchNorm + LeakyReLU + both linear layers) of every forward pass as ONE
# ticket-scheduled launch (disc_deep_fwd_kernel): parity, bitwise against layer by layer, per-shape times, A/B inside the iteration.
# Also: the igemm kernels were refactored into device bodies (codegen changed): the conv op tests and the full model suites.
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c19; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 2400 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_round5.py tests/test_gpu_fullsize.py tests/test_gpu_api_round3.py -x -q -m gpu > $O/tests_model.log 2>&1; tail -4 $O/tests_model.log
for rep in 1 2 3; do for e in "DBM_X=1" "DBM_DISC_DEEP_FUSED=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab.txt 2>&1; cat $O/ab.txt
for e in "DBM_X=1" "DBM_DISC_DEEP_FUSED=0"; do env $e timeout 300 python3 tools/experiments/step_shapes.py "disc_deep|_pm|c128>256|c256>256" 10 2>&1 | grep -E "disc_deep|_pm|c128>256|c256>256|env" | sed "s/^/[$e] /"; done > $O/shapes.txt 2>&1; cat $O/shapes.txt
