#!/bin/bash
# round 6, call 23: the bf16 sweep's 64 -> 64 deformable layer with its sampler on an LDS window (deform_conv64_x3w_kernel): bitwise
# against the gathering kernel, the crop fixture with both, and the crop / continent times with both.
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c23; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -3 $O/tests_ops.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" > $O/tests_config5.log 2>&1; tail -3 $O/tests_config5.log
for rep in 1 2; do for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform 10 2>&1 | tail -4; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
