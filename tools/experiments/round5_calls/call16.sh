#!/bin/bash
# round 5, call 16: the fused input block, early sampling lists, early cleargrads -- parity (model / api / dem / fullsize suites) + A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c16
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_round5.py tests/test_gpu_api_round3.py tests/test_gpu_dem.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15 > gpurun_out/r5c16/tests.txt
bash tools/experiments/ab_env.sh "DBM_X=1" "DBM_INPUT_FUSED=0" "DBM_ITER_CSR_EARLY=0" > gpurun_out/r5c16/ab.txt 2>&1
python3 tools/phases.py fused > gpurun_out/r5c16/phases.txt 2>&1
cat gpurun_out/r5c16/tests.txt gpurun_out/r5c16/ab.txt gpurun_out/r5c16/phases.txt
