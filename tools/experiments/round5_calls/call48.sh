#!/bin/bash
# round 5, call 48: the trunk weight-gradient launch planned for fewer than all 1024 slots (slack for what still holds CUs when it starts)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c48
for rep in 1 2 3; do
  for e in "DBM_X=1" "DBM_WGRAD_SLOTS=960" "DBM_WGRAD_SLOTS=896" "DBM_WGRAD_SLOTS=768" "DBM_WGRAD_SLOTS=640"; do
    echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done > gpurun_out/r5c48/ab.txt 2>&1
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r5c48/ab.txt"):
    m = re.match(r"\[(.*?)\] ms_per_step ([0-9.]+)", l)
    if m: d[m.group(1)].append(float(m.group(2)))
for k, v in d.items(): print("%-32s %s  median %.3f" % (k, " ".join("%.3f" % x for x in v), sorted(v)[len(v) // 2]))
PY
for e in "DBM_X=1" "DBM_WGRAD_SLOTS=896"; do env $e timeout 300 python3 tools/experiments/step_shapes.py wave_dma 10 2>&1 | grep -E "wave_dma_x1[0-9]" | sed "s/^/[$e] /"; done
