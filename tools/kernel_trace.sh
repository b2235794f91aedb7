#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of an arbitrary python script; prints every dispatch whose name contains <filter>.
# usage: tools/kernel_trace.sh <tag> <filter> <script.py> [args...]
set -e
TAG=$1; FILT=$2; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace -f csv -d "$OUT" -o t -- python3 "$@" > "$OUT/log.txt" 2>&1 || { echo "rocprofv3 failed; last lines of $OUT/log.txt:" >&2; tail -n 30 "$OUT/log.txt" >&2; exit 1; }
python3 - "$OUT" "$FILT" <<'PY'
import csv, sys, os
out, filt = sys.argv[1], sys.argv[2]
src = os.path.join(out, "t_kernel_trace.csv")
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(os.path.join(out, "dispatches.txt"), "w") as f:
    for r in rows:
        if filt in r["Kernel_Name"]:
            line = "%-60s %9.1f us grid=%s wg=%s vgpr=%s lds=%s" % (r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                                  r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
            print(line); f.write(line + "\n")
os.remove(src)
PY
