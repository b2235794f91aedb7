#!/bin/bash
# round 6, call 49: the MFMA premultiplication with coalesced loads (a tile transposed into the B layout through LDS): parity, crop A/B
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c49; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or continent" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for rep in 1 2; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_DEFORM1_PREMUL_MFMA=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform1 10 2>&1 | tail -2 | sed 's/.root.repo.deepbedmap_amd.//'; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
