#!/bin/bash
# round 5, call 45: finer deterministic K split of the workgroup-form weight gradients (DBM_WGRAD_DET_MINWG; default 128)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c45
for rep in 1 2 3; do
  for e in "DBM_X=1" "DBM_WGRAD_DET_MINWG=256" "DBM_WGRAD_DET_MINWG=448" "DBM_WGRAD_DET_MINWG=64"; do
    echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done > gpurun_out/r5c45/ab.txt 2>&1
for e in "DBM_X=1" "DBM_WGRAD_DET_MINWG=448"; do env $e timeout 300 python3 tools/experiments/step_shapes.py wgrad 10 2>&1 | grep -E "wgrad<|env" | sed "s/^/[$e] /" >> gpurun_out/r5c45/ab.txt; done
cat gpurun_out/r5c45/ab.txt
