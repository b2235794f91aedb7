#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5c14; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_dem.py -x -q -m gpu > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log
bash tools/experiments/ab_env.sh "DBM_DEFORM1_PREMUL_BWD=0" "DBM_X=1" > $O/ab.txt 2>&1; cat $O/ab.txt
python3 tools/phases.py fused > $O/phases.txt 2>&1; cat $O/phases.txt
