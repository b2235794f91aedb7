#!/bin/bash
# round 5, call 24: everything the second half of round 5 added, switched off together, against the default -- on ONE box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c24
OFF="DBM_CIN_LIVE=0 DBM_CONV_TILE_YT=0 DBM_INPUT_FUSED=0 DBM_ITER_CSR_EARLY=0 DBM_PACK_SPLIT=0 DBM_CONV_TILE_K4=0"
bash tools/experiments/ab_env.sh "DBM_X=1" "$OFF" "DBM_X=1" "$OFF" > gpurun_out/r5c24/ab_late_round5.txt 2>&1
cat gpurun_out/r5c24/ab_late_round5.txt
python3 bench.py --no-cpu-baseline --no-continent --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['extras']['sweep']['bf16']['ms_per_crop'])"
