#!/bin/bash
# round 5, call 21: the stride-2 / 9x9 forms of conv_tile re-measured inside the current iteration; round5 test file
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c21
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -5 > gpurun_out/r5c21/tests.txt
bash tools/experiments/ab_env.sh "DBM_X=1" "DBM_CONV_TILE_K4=1" "DBM_CONV_TILE_9=1" "DBM_CONV_TILE_K4=1 DBM_CONV_TILE_9=1" > gpurun_out/r5c21/ab.txt 2>&1
cat gpurun_out/r5c21/tests.txt gpurun_out/r5c21/ab.txt
