#!/bin/bash
# round 6, call 42: final_conv_layer1's fp32 forward on the training tile with an LDS window of full rows (deform_conv64_fusedw_kernel):
# parity + bitwise against the gathering kernel (switch test), per-shape time, A/B of the training step
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c42; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "schedule_switches and (env10 or env11)" > $O/tests_switch.log 2>&1; tail -2 $O/tests_switch.log
for rep in 1 2 3; do for e in "DBM_X=1" "DBM_DEFORM_FWD_WINDOW=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
for e in "DBM_X=1" "DBM_DEFORM_FWD_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/step_shapes.py "deform" 10 2>&1 | grep -E "deform|env" | sed "s/^/[$e] /"; done > $O/shapes.txt 2>&1; cat $O/shapes.txt
