#!/bin/bash
# round 6, call 30: LDS-window deformable kernel (ablation switch compiled out of the product build): bitwise tests + crop A/B
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c30; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
for rep in 1 2 3; do for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 10 2>&1 | tail -2; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
