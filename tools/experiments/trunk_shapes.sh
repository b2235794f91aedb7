#!/bin/bash
# usage: trunk_shapes.sh "VAR=a" ...: standalone / in-step time of the three persistent trunk launches under each environment
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for e in "$@"; do
  env $e timeout 300 python3 bench.py --no-cpu-baseline --no-sweep --steps 30 --warmup 5 > gpurun_out/ts.json 2> gpurun_out/ts.err || tail -3 gpurun_out/ts.err
  python3 - "$e" <<'PY'
import json, sys
b = json.load(open('gpurun_out/ts.json'))
t = [p for p in b['roofline']['per_shape'] if p['kernel'].startswith('trunk_fused')]
print(sys.argv[1], 'step %.3f ms' % b['ms_per_step'], ' | '.join('%s %.1f us (in-step %.1f)' % (p['shape'], p['avg_us_standalone'], 1e3 * p['ms']) for p in t))
PY
done
