#!/bin/bash
# round 6, call 36: the whole -m gpu suite on the round's final code (LDS-window deformable kernel, premultiplication on MFMA, two
# workgroups per CU for the offset convolutions), then everything profiles/r6/ holds for it (tools/collect_profiles.sh)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c36; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "pytest rc $?" >> $O/full_tests.log; tail -4 $O/full_tests.log
bash tools/collect_profiles.sh r6z2 z_round6 > $O/collect.log 2>&1; tail -25 $O/collect.log
cut -c1-600 gpurun_out/r6z2/z_round6_bench.json
