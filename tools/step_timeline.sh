#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of tools/step_only.py; per-queue timeline of one steady-state iteration ->
# gpurun_out/<tag>/timeline.txt (+ kernel_stats.csv).   usage: tools/step_timeline.sh <tag> [iterations]
set -e
TAG=${1:-tl}; N=${2:-30}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o t -- python3 tools/step_only.py "$N" > "$OUT/log.txt" 2>&1 || { tail -n 30 "$OUT/log.txt" >&2; exit 1; }
grep ms_per_step "$OUT/log.txt" || true
python3 - "$OUT" <<'PY'
import csv, sys, os, re
out = sys.argv[1]
src = os.path.join(out, "t_kernel_trace.csv")
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
# an iteration = (generator adam of the previous one, generator adam of this one]; two adam launches per iteration
k = len(ad) // 2
a0, a1 = ad[k - (k % 2) - 1], ad[k - (k % 2) + 1]
seg = rows[a0 + 1:a1 + 1]
t0 = int(seg[0]["Start_Timestamp"])
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"(\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:40]
with open(os.path.join(out, "timeline.txt"), "w") as f:
    f.write("iteration span %.1f us, %d dispatches\n" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e3, len(seg)))
    for q in sorted(set(r["Queue_Id"] for r in seg)):
        f.write("=== queue %s\n" % q)
        cur = None
        for r in seg:
            if r["Queue_Id"] != q:
                continue
            s, e, n = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, short(r["Kernel_Name"])
            if cur and cur[0] == n and s - cur[2] < 40:
                cur[1] += 1; cur[2] = e; cur[4] += e - s
            else:
                if cur: f.write("  %-42s x%-3d %8.1f -> %8.1f  busy %7.1f\n" % (cur[0], cur[1], cur[3], cur[2], cur[4]))
                cur = [n, 1, e, s, e - s]
        if cur: f.write("  %-42s x%-3d %8.1f -> %8.1f  busy %7.1f\n" % (cur[0], cur[1], cur[3], cur[2], cur[4]))
os.remove(src)
PY
mv "$OUT/t_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null || true
rm -f "$OUT"/t_agent_info.csv "$OUT"/t_domain_stats.csv
cat "$OUT/timeline.txt"
