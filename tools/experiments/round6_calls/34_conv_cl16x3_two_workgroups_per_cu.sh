#!/bin/bash
# round 6, call 34: the split-bf16 convolutions with one output-channel tile (the offset convolutions, 64 -> 18 at full resolution) as
# tiles of <= 8 patches with TWO workgroups per CU: parity tests, crop fixture, A/B (DBM_CL16X3_PAIR=0, libdbm_measure.so)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c34; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
timeout 900 python3 -m pytest tests/test_gpu_cl16.py -x -q -m gpu > $O/tests_cl16.log 2>&1; tail -2 $O/tests_cl16.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or continent" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for rep in 1 2; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_CL16X3_PAIR=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py "x3_c64>18" 10 2>&1 | tail -2 | sed 's/.root.repo.deepbedmap_amd.//'; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
