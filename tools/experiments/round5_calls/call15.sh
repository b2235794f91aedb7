#!/bin/bash
# round 5, call 15: parity of the two conv_tile trims (gradient-channel skip, channels-last twin) + A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c15
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_round5.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15 > gpurun_out/r5c15/tests.txt
bash tools/experiments/ab_env.sh "DBM_X=1" "DBM_CIN_LIVE=0" "DBM_CONV_TILE_YT=0" "DBM_CIN_LIVE=0 DBM_CONV_TILE_YT=0" > gpurun_out/r5c15/ab.txt 2>&1
cat gpurun_out/r5c15/tests.txt gpurun_out/r5c15/ab.txt
