#!/bin/bash
# round 5, call 49: the input block on large planes as one launch (no im2col image) -- parity + the sweep crop
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c49
{
timeout 2400 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_dem.py tests/test_gpu_cl16.py -x -q 2>&1 | tail -3
for e in "DBM_X=1" "DBM_INPUT_FUSED=0" "DBM_X=1" "DBM_INPUT_FUSED=0"; do
  echo "[$e] $(env $e timeout 300 python3 tools/sweep_crop_bench.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32', round(d['fp32']['ms_per_crop'],3), 'bf16', round(d['bf16']['ms_per_crop'],3), 'batch8', round(d['bf16'].get('batch8_ms_per_crop',0),3))")"
done
} > gpurun_out/r5c49/out.txt 2>&1
cat gpurun_out/r5c49/out.txt
