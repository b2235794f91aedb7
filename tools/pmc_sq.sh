#!/bin/bash
# Runs on the GPU box: SQ-level counters (MFMA busy, wave wait states) per kernel, one pass.
# usage: tools/pmc_sq.sh <tag>            (SQ_CMD="tools/sweep_crop_bench.py" tools/pmc_sq.sh <tag>: the sweep's kernels instead of the training step's)
set -e
TAG=${1:-pmcsq}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv -d "$OUT" -o sq -- python3 ${SQ_CMD:-bench.py --no-cpu-baseline --no-sweep --tables /tmp/sq_tables.json --steps 1 --warmup 1} > "$OUT/sq.log" 2>&1 || { echo "rocprofv3 failed; last lines of "$OUT/sq.log":" >&2; tail -n 30 "$OUT/sq.log" >&2; exit 1; }
python3 - "$OUT" <<'PY'
import csv, json, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f"{out}/sq_counter_collection.csv")):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
res = {k: dict(v, launches=cnt[k]) for k, v in acc.items()}
for k, v in res.items():
    # MFMA-busy %: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (it equals 64 cycles x the kernel's
    # v_mfma_f32_32x32x2_f32 count, 32 x its v_mfma_f32_32x32x16_bf16 count), GRBM_GUI_ACTIVE over the 8 XCDs: the denominator is
    # (GRBM_GUI_ACTIVE / 8) x 1024 SIMD-cycles = GRBM_GUI_ACTIVE x 128 -- every SIMD issuing MFMAs back to back for the whole
    # (serialised) launch = 100 %
    if v.get("GRBM_GUI_ACTIVE"):
        v["mfma_busy_pct"] = 100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["GRBM_GUI_ACTIVE"] * 128.0)
        v["mfma_busy_denominator"] = "GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs"
        # wavefront occupancy in the same units: SQ_WAVE_CYCLES counts quad-cycles per resident wavefront
        v["avg_waves_per_simd"] = 4.0 * v.get("SQ_WAVE_CYCLES", 0.0) / (v["GRBM_GUI_ACTIVE"] * 128.0)
json.dump(res, open(f"{out}/sq.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a != "mfma_busy_denominator"})
PY
rm -f "$OUT"/*_kernel_trace.csv "$OUT"/*_counter_collection.csv
