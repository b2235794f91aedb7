#!/bin/bash
# the data-parallel schedule on one GPU (bench.py --force-comm): step time and the weight-gradient launch shapes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for e in "$@"; do
  env $e timeout 300 python3 bench.py --force-comm --no-cpu-baseline --no-sweep --steps 60 --warmup 5 > gpurun_out/dp.json 2> gpurun_out/dp.err || tail -3 gpurun_out/dp.err
  python3 - "$e" <<'PY'
import json, sys
b = json.load(open('gpurun_out/dp.json'))
print(sys.argv[1], 'step %.3f ms' % b['ms_per_step'], b['config'].get('gradient_exchange'))
for p in b['roofline']['per_shape']:
    if p['kernel'] in ('wgrad_kernel', 'trunk_fused_kernel', 'trunk_fused_bwd_kernel') and p['ms_standalone'] > 0.08:
        print('   %-26s wg %5d n %d standalone %.1f us in-step %.3f ms frac %.3f' % (p['shape'], p['workgroups'], p['launches'], p['avg_us_standalone'], p['ms'], p['frac_mfma_standalone']))
PY
done
