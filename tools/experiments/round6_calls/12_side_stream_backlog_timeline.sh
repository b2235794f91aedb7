#!/bin/bash
# round 6, call 12: phase marks around the trunk's weight-gradient launch on the side stream (how long after the chain's end does it start?)
# and the per-queue timeline of one iteration on the current code
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c12; mkdir -p $O
python3 tools/phases.py fused > $O/phases.txt 2>&1; cat $O/phases.txt
bash tools/step_timeline.sh r6c12_tl > $O/timeline.log 2>&1; tail -5 $O/timeline.log
ls gpurun_out/r6c12_tl 2>/dev/null | head
