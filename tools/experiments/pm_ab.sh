#!/bin/bash
# A/B of the position-major deep-discriminator convolutions (DBM_IGEMM_PM) inside the training step + standalone per shape
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for rep in 1 2; do
  for pm in 0 1; do
    echo "PM=$pm $(DBM_IGEMM_PM=$pm timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done
for kt in 256 512 2048; do
  echo "PM=1 KSTARGET=$kt $(DBM_IGEMM_PM_KSTARGET=$kt timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
done
for pm in 0 1; do
  DBM_IGEMM_PM=$pm timeout 300 python3 bench.py --no-cpu-baseline --no-sweep --steps 60 --warmup 5 > gpurun_out/pm_bench_$pm.json 2> gpurun_out/pm_bench_$pm.err
done
python3 - <<'PY'
import json
for pm in (0, 1):
    try:
        b = json.load(open('gpurun_out/pm_bench_%d.json' % pm))
    except Exception as e:
        print(pm, 'bench failed', e); continue
    print('PM', pm, 'ms/step', b['ms_per_step'], 'igemm frac_standalone', b['roofline']['frac_standalone'], 'frac', b['roofline']['frac'])
    for p in b['roofline']['per_shape']:
        s = p['shape']
        if p['kernel'] == 'igemm_conv_kernel' and any(k in s for k in ('_2x2', '_4x4', '_1x1')):
            print('   %-26s wg %5d n %d in-step %.3f ms standalone %.1f us each' % (s, p['workgroups'], p['launches'], p['ms'], p['avg_us_standalone']))
PY
