#!/bin/bash
# like wgrad_shapes.sh, every weight-gradient shape
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for e in "$@"; do
  env $e timeout 300 python3 bench.py --no-cpu-baseline --no-sweep --steps 40 --warmup 5 > gpurun_out/ws.json 2> gpurun_out/ws.err
  python3 - "$e" <<'PY'
import json, sys
b = json.load(open('gpurun_out/ws.json'))
ws = [p for p in b['roofline']['per_shape'] if p['kernel'] == 'wgrad_kernel']
print(sys.argv[1], 'step %.3f ms' % b['ms_per_step'], 'wgrad standalone %.3f ms in-step %.3f ms' % (sum(p['ms_standalone'] for p in ws), sum(p['ms'] for p in ws)))
for p in ws:
    print('   %-18s wg %5d standalone %.1f us in-step %.3f ms gf %.2f frac %.3f' % (p['shape'], p['workgroups'], p['avg_us_standalone'], p['ms'], p['gflop_per_launch'], p['frac_mfma_standalone']))
PY
done
