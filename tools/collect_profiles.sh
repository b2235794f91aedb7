#!/bin/bash
# Runs on the GPU box: everything profiles/<round>/ holds for one state of the code, into gpurun_out/<tag>/ under the names the
# round directory uses.  usage: tools/collect_profiles.sh <tag> <prefix>      (e.g. r3z z_round3_final)
#   <prefix>_bench.json                     default `python bench.py` (unprofiled; cpu_baseline + extras.sweep included)
#   <prefix>_bench_tables.json              the per-shape tables OF THAT RUN (bench.py --tables)
#   <prefix>_bench_profiled.json            the line of the same command under rocprofv3 --kernel-trace --stats
#   <prefix>_kernel_stats.csv               rocprofv3 --stats per-kernel summary (in-step: streams overlap)
#   <prefix>_serialised_kernel_stats.csv    the same with --pmc GRBM_GUI_ACTIVE (dispatches serialised: standalone durations)
#   <prefix>_phases.txt                     tools/phases.py fused (hipEvents at the phase boundaries, unprofiled)
#   <prefix>_timeline_per_queue.txt         tools/step_timeline.sh (per-queue run-length summary of one iteration)
#   traffic_pmc.json                        tools/pmc_traffic.sh (FETCH_SIZE / WRITE_SIZE passes; per family and per launch shape)
#   sq_counters.json                        tools/pmc_sq.sh (MFMA-busy %, wait states per kernel: the training step)
#   sq_counters_sweep.json                  the same for the sweep's kernels (tools/sweep_crop_bench.py)
set -e
TAG=${1:?tag}; PFX=${2:?prefix}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
# (the per-shape tables of THIS unprofiled run are written straight to their final name: the profiled passes below write their own
#  tables elsewhere -- round 4 copied bench_tables.json after the --pmc passes had overwritten it)
python3 bench.py --tables "$OUT/${PFX}_bench_tables.json" > "$OUT/${PFX}_bench.json" 2> "$OUT/bench.err" || { tail -n 20 "$OUT/bench.err" >&2; exit 1; }
bash tools/profile_bench.sh $TAG/prof > /dev/null
cp "$OUT/prof/bench.json" "$OUT/${PFX}_bench_profiled.json"
cp "$OUT/prof/trace_kernel_stats.csv" "$OUT/${PFX}_kernel_stats.csv"
bash tools/serial_kernel_times.sh $TAG/ser > /dev/null
cp "$OUT/ser/ser_kernel_stats.csv" "$OUT/${PFX}_serialised_kernel_stats.csv"
python3 tools/phases.py fused > "$OUT/${PFX}_phases.txt" 2>&1 || true
bash tools/step_timeline.sh $TAG/tl 30 > /dev/null
cp "$OUT/tl/timeline.txt" "$OUT/${PFX}_timeline_per_queue.txt"
bash tools/pmc_traffic.sh $TAG/pmc > "$OUT/pmc_traffic.log"
cp "$OUT/pmc/traffic.json" "$OUT/traffic_pmc.json"
bash tools/pmc_sq.sh $TAG/sq > "$OUT/pmc_sq.log"
cp "$OUT/sq/sq.json" "$OUT/sq_counters.json"
SQ_CMD="tools/sweep_crop_bench.py" bash tools/pmc_sq.sh $TAG/sqs > "$OUT/pmc_sq_sweep.log"
cp "$OUT/sqs/sq.json" "$OUT/sq_counters_sweep.json"
rm -rf "$OUT/prof" "$OUT/ser" "$OUT/tl" "$OUT/pmc" "$OUT/sq" "$OUT/sqs"
ls -la "$OUT"
