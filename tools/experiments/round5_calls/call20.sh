#!/bin/bash
# round 5, call 20: forward / data-gradient weight images packed separately -- whole GPU suite + A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c20
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r5c20/tests.txt
bash tools/experiments/ab_env.sh "DBM_X=1" "DBM_PACK_SPLIT=0" > gpurun_out/r5c20/ab.txt 2>&1
cat gpurun_out/r5c20/tests.txt gpurun_out/r5c20/ab.txt
