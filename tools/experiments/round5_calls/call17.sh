#!/bin/bash
# round 5, call 17: where the early sampling-list build goes (side stream / chain[0] / inside the backward pass), fused input block A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c17
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -8 > gpurun_out/r5c17/tests.txt
bash tools/experiments/ab_env.sh "DBM_ITER_CSR_EARLY=0" "DBM_ITER_CSR_EARLY=1" "DBM_ITER_CSR_EARLY=2" "DBM_ITER_CSR_EARLY=0 DBM_INPUT_FUSED=0" > gpurun_out/r5c17/ab.txt 2>&1
DBM_ITER_CSR_EARLY=0 python3 tools/phases.py fused > gpurun_out/r5c17/phases_csr0.txt 2>&1
DBM_ITER_CSR_EARLY=2 python3 tools/phases.py fused > gpurun_out/r5c17/phases_csr2.txt 2>&1
cat gpurun_out/r5c17/tests.txt gpurun_out/r5c17/ab.txt gpurun_out/r5c17/phases_csr0.txt gpurun_out/r5c17/phases_csr2.txt
