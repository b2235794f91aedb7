#!/bin/bash
# Runs on the GPU box: per-kernel STANDALONE durations (rocprofv3 --pmc serialises the dispatches: no two kernels overlap),
# i.e. the chip time each kernel family costs per training step when nothing shares the GPU with it.
# usage: tools/serial_kernel_times.sh <tag>
set -e
TAG=${1:-serial}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --stats -f csv -d "$OUT" -o ser -- python3 bench.py --no-cpu-baseline --no-continent --tables "$OUT/tables.json" --steps 4 --warmup 1 > "$OUT/bench.log" 2>&1 || { echo "rocprofv3 failed; last lines of "$OUT/bench.log":" >&2; tail -n 30 "$OUT/bench.log" >&2; exit 1; }
rm -f "$OUT"/ser_kernel_trace.csv "$OUT"/ser_counter_collection.csv
ls "$OUT"
