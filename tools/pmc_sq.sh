#!/bin/bash
# Runs on the GPU box: SQ-level counters (MFMA busy, wave wait states) per kernel, one pass.
# usage: tools/pmc_sq.sh <tag>
set -e
TAG=${1:-pmcsq}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv -d "$OUT" -o sq -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > "$OUT/sq.log" 2>&1 || { echo "rocprofv3 failed; last lines of "$OUT/sq.log":" >&2; tail -n 30 "$OUT/sq.log" >&2; exit 1; }
python3 - "$OUT" <<'PY'
import csv, json, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f"{out}/sq_counter_collection.csv")):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
res = {k: dict(v, launches=cnt[k]) for k, v in acc.items()}
json.dump(res, open(f"{out}/sq.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print(k, {a: round(b) for a, b in v.items()})
PY
rm -f "$OUT"/*_kernel_trace.csv "$OUT"/*_counter_collection.csv
