#!/bin/bash
# round 6, call 40: conv_layer5 of every dense block (192 -> 64) in the two-workgroup form too -- two workgroups of 32 output channels per
# tile: tests, continent A/B (DBM_CL16_PAIR=0 switches both off; libdbm_measure.so)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c40; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
timeout 900 python3 -m pytest tests/test_gpu_cl16.py -x -q -m gpu > $O/tests_cl16.log 2>&1; tail -2 $O/tests_cl16.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or continent" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for rep in 1 2; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_CL16_PAIR=0"; do echo "[${e#*so}] $(env $e timeout 600 python3 tools/continent_sweep.py 8 2>&1 | tail -1 | grep -o '"sweep_s": [0-9.]*, "ms_per_tile": [0-9.]*' | head -1)"; done; done > $O/ab_continent.txt 2>&1; cat $O/ab_continent.txt
