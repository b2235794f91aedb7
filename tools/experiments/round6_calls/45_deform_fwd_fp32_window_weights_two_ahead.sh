#!/bin/bash
# round 6, call 45: deform_conv64_fusedw_kernel with the weights requested two steps ahead (four buffers, one wait count) and both window
# kernels' corner requests as ONE asm statement: parity (ops with and without DBM_DEFORM_FWD_WINDOW=1, bitwise switch tests, crop fixtures),
# per-shape times, step and crop A/B
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c45; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
DBM_DEFORM_FWD_WINDOW=1 timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops_fwin.log 2>&1; tail -2 $O/tests_ops_fwin.log
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "schedule_switches and (env10 or env11)" > $O/tests_switch.log 2>&1; tail -2 $O/tests_switch.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or continent" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for rep in 1 2 3; do for e in "DBM_DEFORM_FWD_WINDOW=1" "DBM_DEFORM_FWD_WINDOW=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
for a in 0 256 1; do echo "abl $a: $(DBM_LIB=$M DBM_DEFORM_FWD_WINDOW=1 DBM_FUSEDW_ABL=$a timeout 300 python3 tools/experiments/step_shapes.py deform64 5 2>&1 | grep deform64 | head -1)"; done > $O/abl.txt 2>&1; cat $O/abl.txt
for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 10 2>&1 | tail -2; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
