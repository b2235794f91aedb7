#!/bin/bash
# round 5, call 25: each late-round-5 switch off by itself against the default, three alternations on one box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c25
for rep in 1 2 3; do
  for e in "DBM_X=1" "DBM_CIN_LIVE=0" "DBM_CONV_TILE_YT=0" "DBM_INPUT_FUSED=0" "DBM_ITER_CSR_EARLY=0" "DBM_PACK_SPLIT=0" "DBM_CONV_TILE_K4=0" "DBM_CONV_TILE=0"; do
    echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done > gpurun_out/r5c25/ab_each.txt 2>&1
cat gpurun_out/r5c25/ab_each.txt
