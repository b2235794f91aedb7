#!/bin/bash
# round 5, first GPU call: the ADVICE regressions, the schedule knob DBM_ITER_EARLY_TWIN, and VERDICT r4 item 1(a):
# what the discriminator's work / the trunk weight gradients cost INSIDE the iteration (libdbm_measure.so, DBM_ITER_ABL)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r5c1; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_api_round3.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
bash tools/experiments/ab_env.sh "DBM_X=0" "DBM_ITER_EARLY_TWIN=1" "DBM_ITER_EARLY_TWIN=2" "DBM_ITER_EARLY_TWIN=2 DBM_TRUNK_HELPER=0" "DBM_ITER_EARLY_TWIN=1 DBM_TRUNK_HELPER=3" > $O/ab_twin.txt 2>&1
cat $O/ab_twin.txt
M=$PWD/deepbedmap_amd/libdbm_measure.so
bash tools/experiments/ab_env.sh "DBM_LIB=$M" "DBM_LIB=$M DBM_ITER_ABL=1" "DBM_LIB=$M DBM_ITER_ABL=2" "DBM_LIB=$M DBM_ITER_ABL=4" "DBM_LIB=$M DBM_ITER_ABL=8" "DBM_LIB=$M DBM_ITER_ABL=3" "DBM_LIB=$M DBM_ITER_ABL=6" "DBM_LIB=$M DBM_ITER_ABL=7" > $O/ab_abl.txt 2>&1
cat $O/ab_abl.txt
python3 tools/phases.py fused > $O/phases_base.txt 2>&1
DBM_ITER_EARLY_TWIN=1 python3 tools/phases.py fused > $O/phases_early1.txt 2>&1
DBM_ITER_EARLY_TWIN=2 python3 tools/phases.py fused > $O/phases_early2.txt 2>&1
DBM_LIB=$M DBM_ITER_ABL=1 python3 tools/phases.py fused > $O/phases_noD.txt 2>&1
