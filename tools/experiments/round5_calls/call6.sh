#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5c6; mkdir -p $O
timeout 2700 python3 -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; tail -5 $O/full_tests.log
python3 tools/sweep_crop_bench.py > $O/sweep.txt 2>&1; tail -30 $O/sweep.txt | cut -c1-400
bash tools/experiments/ab_env.sh "DBM_CONV_TILE=0" "DBM_CONV_TILE=1" > $O/ab.txt 2>&1; cat $O/ab.txt
