#!/bin/bash
# round 5, call 22: deformable 64 -> 64 backward with late stores -- parity + standalone time + step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c22
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_round5.py tests/test_gpu_dem.py -x -q 2>&1 | tail -6 > gpurun_out/r5c22/tests.txt
timeout 300 python3 tools/experiments/step_shapes.py deform 30 > gpurun_out/r5c22/shapes.txt 2>&1
bash tools/experiments/ab_env.sh "DBM_X=1" "DBM_X=2" > gpurun_out/r5c22/ab.txt 2>&1
cat gpurun_out/r5c22/tests.txt gpurun_out/r5c22/shapes.txt gpurun_out/r5c22/ab.txt
