#!/bin/bash
# round 5, call 47: the per-layer convolution's global launch knobs + two schedule knobs re-measured on the final schedule
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c47
for rep in 1 2 3; do
  for e in "DBM_X=1" "DBM_IGEMM_PM_KSTARGET=256" "DBM_IGEMM_PM_KSTARGET=1024" "DBM_IGEMM_KSTARGET=128" "DBM_IGEMM_KSTARGET=512" "DBM_IGEMM_W4_TILES=512" "DBM_IGEMM_W4_TILES=2048" "DBM_IGEMM_W8_TILES=256" "DBM_IGEMM_W8_TILES=1024" "DBM_IGEMM_MINPAIRS=2" "DBM_IGEMM_MINPAIRS=6" "DBM_IGEMM_MT2_TILES=1024" "DBM_ITER_AUX=1" "DBM_TRUNK_LOCAL_ST=0"; do
    echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done > gpurun_out/r5c47/ab.txt 2>&1
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r5c47/ab.txt"):
    m = re.match(r"\[(.*?)\] ms_per_step ([0-9.]+)", l)
    if m: d[m.group(1)].append(float(m.group(2)))
for k, v in d.items(): print("%-32s %s  median %.3f" % (k, " ".join("%.3f" % x for x in v), sorted(v)[len(v) // 2]))
PY
