#!/bin/bash
# round 5, call 52: the sweep pre- and post-residual convolutions in split-bf16 on NHWC (no layout conversions) -- parity / accuracy + crop time
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c52
{
timeout 2400 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_dem.py tests/test_gpu_cl16.py -x -q 2>&1 | tail -3
for e in "DBM_X=1" "DBM_PRE_X3=0" "DBM_X=1" "DBM_PRE_X3=0" "DBM_POST_X3=0"; do
  echo "[$e] $(env $e timeout 300 python3 tools/sweep_crop_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', round(d['fp32']['ms_per_crop'],3), 'bf16', round(d['bf16']['ms_per_crop'],3))")"
done
timeout 600 python3 tools/bf16_error_study.py 2>&1 | tail -6
} > gpurun_out/r5c52/out.txt 2>&1
cat gpurun_out/r5c52/out.txt
