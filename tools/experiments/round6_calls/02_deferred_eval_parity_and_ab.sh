#!/bin/bash
# round 6, call 2: the G-step's eval-mode discriminator pass deferred to the next library call (DBM_ITER_DEFER_EVAL, default 1):
# parity (bitwise against =0; timeouts; soak; data-parallel one-rank schedule), then A/B of the switch and of the placement of the
# deferred pass inside the next iteration (tuning switches of libdbm_measure.so: DBM_ITER_DEFER_AT, DBM_ITER_DEFER_PACK)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c2; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_api_round3.py tests/test_gpu_parallel.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -6 $O/tests.log
timeout 1200 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "timeout or trainer or epoch or iteration or minibatch or step" > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
bash tools/experiments/ab_env.sh "DBM_X=1" "DBM_ITER_DEFER_EVAL=0" > $O/ab_defer.txt 2>&1; cat $O/ab_defer.txt
M=$PWD/deepbedmap_amd/libdbm_measure.so
bash tools/experiments/ab_env.sh "DBM_LIB=$M" "DBM_LIB=$M DBM_ITER_DEFER_AT=0" "DBM_LIB=$M DBM_ITER_DEFER_AT=1" "DBM_LIB=$M DBM_ITER_DEFER_AT=3" "DBM_LIB=$M DBM_ITER_DEFER_PACK=0" "DBM_LIB=$M DBM_ITER_DEFER_AT=3 DBM_ITER_DEFER_PACK=0" > $O/ab_defer_at.txt 2>&1; cat $O/ab_defer_at.txt
python3 tools/phases.py fused > $O/phases.txt 2>&1; cat $O/phases.txt
