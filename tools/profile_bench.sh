#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of the default bench command; summaries land in gpurun_out/<tag>/
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-prof}; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o trace -- python3 bench.py --no-cpu-baseline --no-continent --tables "$OUT/tables.json" "$@" > "$OUT/bench.log" 2>&1 || { echo "rocprofv3 failed; last lines of "$OUT/bench.log":" >&2; tail -n 30 "$OUT/bench.log" >&2; exit 1; }
grep '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
# per-dispatch trace is large: keep only the last 2600 dispatches (about two training step) for inspection
python3 - "$OUT" <<'PY'
import csv, sys, os
out = sys.argv[1]
src = os.path.join(out, "trace_kernel_trace.csv")
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-2600:]
with open(os.path.join(out, "last_step_dispatches.csv"), "w") as f:
    t0 = int(tail[0]["Start_Timestamp"])
    f.write("kernel,start_us,duration_us,queue,grid,workgroup,lds,vgpr\n")
    for r in tail:
        f.write('"%s",%.2f,%.2f,%s,%s,%s,%s,%s\n' % (r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", ""),
                                           r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                                           r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
os.remove(src)
PY
ls "$OUT"
