#!/bin/bash
# round 5, call 46: the weight-gradient launch-size knobs re-measured on the final schedule
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c46
for rep in 1 2 3; do
  for e in "DBM_X=1" "DBM_WGRAD_DIRECT_WGS=256" "DBM_WGRAD_DIRECT_WGS=768" "DBM_WGRAD_DIRECT_WGS=1024" "DBM_WGRAD_1X1_WGS=256" "DBM_WGRAD_1X1_WGS=1024" "DBM_WGRAD_SLOTS_SMALL=128" "DBM_WGRAD_SLOTS_SMALL=512" "DBM_WGRAD_SLOTS=512" "DBM_DBWD_ORDER=0"; do
    echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done > gpurun_out/r5c46/ab.txt 2>&1
cat gpurun_out/r5c46/ab.txt
