#!/bin/bash
# round 6, call 1: the whole -m gpu suite on the hygiene commit (ADVICE r5 fixes, tuning switches moved to libdbm_measure.so, the new
# batch-64 summation-order test, the explicit LeakyReLU-flip confirmation) + this round's baseline of the iteration (step_only x3, phases)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c1; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "pytest rc $?" >> $O/full_tests.log
tail -15 $O/full_tests.log
grep -h "batch-64 summation" $O/full_tests.log
for i in 1 2 3; do timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1; done > $O/step_only.txt; cat $O/step_only.txt
python3 tools/phases.py fused > $O/phases.txt 2>&1; cat $O/phases.txt
