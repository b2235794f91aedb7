#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5c7; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv2d" > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
DBM_CONV_TILE_K4=0 DBM_CONV_TILE_9=0 python3 tools/experiments/step_shapes.py igemm 30 > $O/shapes_off.txt 2>&1
python3 tools/experiments/step_shapes.py igemm 30 > $O/shapes_on.txt 2>&1
head -1 $O/shapes_off.txt; head -1 $O/shapes_on.txt
bash tools/experiments/ab_env.sh "DBM_CONV_TILE=0" "DBM_CONV_TILE_K4=0 DBM_CONV_TILE_9=0" "DBM_X=1" > $O/ab.txt 2>&1
cat $O/ab.txt
