#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of the default bench command; summaries land in gpurun_out/<tag>/
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o trace -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
grep '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
rm -f "$OUT"/*kernel_trace.csv   # per-dispatch trace is large; the stats summary is what gets committed
ls "$OUT"
