#!/bin/bash
# round 5, call 26: 32-column trunk tiles (162 workgroups) re-measured on the final round-5 schedule
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c26
bash tools/experiments/ab_env.sh "DBM_X=1" "DBM_TRUNK_TP=32" > gpurun_out/r5c26/ab.txt 2>&1
cat gpurun_out/r5c26/ab.txt
