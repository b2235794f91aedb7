#!/bin/bash
# round 6, call 50: the LDS-window deformable kernel's channels-last output through LDS (whole 256-byte pixels per store): bitwise tests,
# crop fixtures, crop and continent A/B
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c50; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or continent" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for rep in 1 2; do for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 10 2>&1 | tail -2; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do echo "[$e] $(env $e timeout 600 python3 tools/continent_sweep.py 8 2>&1 | tail -1 | grep -o '"sweep_s": [0-9.]*, "ms_per_tile": [0-9.]*' | head -1)"; done > $O/ab_continent.txt 2>&1; cat $O/ab_continent.txt
