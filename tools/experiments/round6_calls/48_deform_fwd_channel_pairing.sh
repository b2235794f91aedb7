#!/bin/bash
# round 6, call 48: final_conv_layer1's fp32 forward with the channel pairing (c, c + 4) per MFMA in BOTH kernels (the window form then
# reads one 16-byte piece per corner and step): the whole -m gpu suite (the tight fixtures see another rounding), per-shape time, step A/B
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c48; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "pytest rc $?" >> $O/full_tests.log; tail -4 $O/full_tests.log
for rep in 1 2 3; do for e in "DBM_X=1" "DBM_DEFORM_FWD_WINDOW=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
for e in "DBM_X=1" "DBM_DEFORM_FWD_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/step_shapes.py "deform64" 10 2>&1 | grep -E "deform64|env" | sed "s/^/[$e] /"; done > $O/shapes.txt 2>&1; cat $O/shapes.txt
