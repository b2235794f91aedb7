#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r5c2; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_api_round3.py -x -q -m gpu -k timeouts > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
bash tools/experiments/ab_env.sh "DBM_X=0" "DBM_ITER_EARLY_TWIN=2" "DBM_ITER_EARLY_TWIN=2 DBM_TRUNK_HELPER=0" "DBM_ITER_EARLY_TWIN=2 DBM_TRUNK_HELPER=3" "DBM_ITER_WGRAD_INLINE=1" "DBM_ITER_EARLY_TWIN=1 DBM_ITER_WGRAD_INLINE=1" > $O/ab_twin.txt 2>&1
cat $O/ab_twin.txt
DBM_ITER_EARLY_TWIN=2 python3 tools/phases.py fused > $O/phases_early2.txt 2>&1
DBM_ITER_EARLY_TWIN=2 DBM_TRUNK_HELPER=0 python3 tools/phases.py fused > $O/phases_early2_h0.txt 2>&1
