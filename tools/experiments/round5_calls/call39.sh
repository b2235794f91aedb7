#!/bin/bash
# round 5, call 39: register-resident branch-free BatchNorm kernels (DBM_BN_REG=0: the general kernels) -- parity, step, standalone times
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c39
{
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_round5.py tests/test_gpu_dem.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for e in "DBM_X=1" "DBM_BN_REG=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done
done
} > gpurun_out/r5c39/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in 1 0; do
  export DBM_BN_REG=$v
  rocprofv3 --kernel-trace --stats --pmc GRBM_GUI_ACTIVE -f csv -d gpurun_out/r5c39/p$v -o p -- python3 tools/step_only.py 10 > /dev/null 2>&1
  f=$(find gpurun_out/r5c39/p$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $v >> gpurun_out/r5c39/bn_serial.txt <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r and 'bn_train' in r[0]:
        print("BN_REG=%s %-36s calls %4s avg %8.1f us  min %7.1f  max %7.1f" % (sys.argv[2], r[0][5:40], r[1], float(r[3]) / 1e3, float(r[5]) / 1e3, float(r[6]) / 1e3))
PY
  rm -rf gpurun_out/r5c39/p$v
done
cat gpurun_out/r5c39/ab.txt gpurun_out/r5c39/bn_serial.txt
