#!/bin/bash
# round 6, call 7: three launch trims on the discriminator's serial chains -- the head as one launch (second version of the forward kernel),
# conv_layer0's weight gradient reading the caller's images (no copy launch), the D-step's cleargrads at the head of the side stream
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c7; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_round5.py tests/test_gpu_api_round3.py -x -q -m gpu -k "discriminator or iteration or soak or schedule or timeout or step" > $O/tests.log 2>&1; tail -3 $O/tests.log
M=$PWD/deepbedmap_amd/libdbm_measure.so
for rep in 1 2 3; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_DISC_HEAD_FUSED=0" "DBM_LIB=$M DBM_DISC_BORROW=0" "DBM_LIB=$M DBM_ITER_DCLEAR_EARLY=0" "DBM_LIB=$M DBM_DISC_HEAD_FUSED=0 DBM_DISC_BORROW=0 DBM_ITER_DCLEAR_EARLY=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_trims.txt 2>&1; cat $O/ab_trims.txt
