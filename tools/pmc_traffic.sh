#!/bin/bash
# Runs on the GPU box: HBM traffic of the dominant kernel from PMC counters, one counter per pass (MI355X_MICROARCH.md
# "HBM": FETCH_SIZE / WRITE_SIZE are in KiB, separate passes; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x).
# usage: tools/pmc_traffic.sh <tag>
set -e
TAG=${1:-pmc}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d "$OUT" -o $C -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/$C.log" 2>&1 || { echo "rocprofv3 failed; last lines of "$OUT/$C.log":" >&2; tail -n 30 "$OUT/$C.log" >&2; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"{out}/{c}_counter_collection.csv")):
        if r["Counter_Name"] != c:
            continue
        kn = r["Kernel_Name"]
        k = ("igemm_conv_kernel" if ("igemm_conv_kernel" in kn or "deform_conv64_fused" in kn or "deform_bwd64_fused" in kn)
             else "trunk_fused_bwd_kernel" if "trunk_fused_bwd_kernel" in kn
             else "trunk_fused_kernel_helper" if "trunk_fused_kernel<27, true>" in kn  # the form with a helper workgroup per image
             else "trunk_fused_kernel" if "trunk_fused_kernel" in kn and "pack" not in kn else "wgrad_kernel" if "wgrad_" in kn else None)
        if k:
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (s, n) in acc.items():
        res.setdefault(k, {})[c + "_KiB_per_launch"] = s / n
        res[k]["launches_sampled"] = n
for k, v in res.items():
    # gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B -> x2 (upper bound for partially coalesced gathers)
    v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE_KiB_per_launch"] + v["WRITE_SIZE_KiB_per_launch"]) * 1024.0
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -f "$OUT"/*_kernel_trace.csv "$OUT"/*_counter_collection.csv
