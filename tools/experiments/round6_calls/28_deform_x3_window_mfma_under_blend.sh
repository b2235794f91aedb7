#!/bin/bash
# round 6, call 28: the LDS-window deformable kernel with the blend in lock-stepped channel pairs and the previous step's MFMAs dealt
# between its stages: bitwise tests, crop A/B, loop ablations
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c28; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for rep in 1 2; do for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 10 2>&1 | tail -2; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
for a in 0 1 14 46 110; do echo "abl $a: $(DBM_LIB=$M DBM_X3W_ABL=$a timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 5 2>&1 | tail -1)"; done > $O/abl.txt 2>&1; cat $O/abl.txt
