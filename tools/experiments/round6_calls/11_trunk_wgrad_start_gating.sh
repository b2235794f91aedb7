#!/bin/bash
# round 6, call 11: what gates the START of the trunk's weight-gradient launch -- the side stream's backlog (tail weight gradients, then two
# im2col launches) ends ~150 us after the chain.  Tuning switches (libdbm_measure.so): DBM_ITER_TRUNKWG_CHAIN0=1 (the launch on chain[0],
# gated by the chain's end alone), DBM_ITER_IM2COL_C0=1 (the im2col images on chain[0], early)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c11; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for rep in 1 2 3; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_ITER_IM2COL_C0=1" "DBM_LIB=$M DBM_ITER_TRUNKWG_CHAIN0=1" "DBM_LIB=$M DBM_ITER_TRUNKWG_CHAIN0=1 DBM_ITER_IM2COL_C0=1"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab.txt 2>&1; cat $O/ab.txt
DBM_LIB=$M DBM_ITER_IM2COL_C0=1 timeout 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config3" > $O/tests.log 2>&1; tail -2 $O/tests.log
