#!/bin/bash
# usage: sweep_ab.sh "VAR=a" "VAR=b" ...: one 288 x 288 crop of the sweep (fp32 and bf16 ms per crop) under each environment
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for e in "$@"; do
  env $e timeout 300 python3 tools/sweep_crop_bench.py > gpurun_out/sw.json 2> gpurun_out/sw.err || tail -3 gpurun_out/sw.err
  python3 - "$e" <<'PY'
import json, sys
s = json.load(open('gpurun_out/sw.json'))
print(sys.argv[1], 'fp32 %.3f ms  bf16 %.3f ms (8 per forward: %s)' % (s['fp32']['ms_per_crop'], s['bf16']['ms_per_crop'], s['bf16'].get('batch8', {}).get('ms_per_crop')))
PY
done
