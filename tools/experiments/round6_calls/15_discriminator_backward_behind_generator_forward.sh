#!/bin/bash
# round 6, call 15: the G-step's own forward tail takes 0.82 ms beside the discriminator's backward passes against 0.36 alone (phase marks).
# DBM_ITER_DBWD_LATE (libdbm_measure.so): 1 = both backward passes behind that forward, 2 = only the fake-batch pass
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c16; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for rep in 1 2 3; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_ITER_DBWD_LATE=3" "DBM_LIB=$M DBM_ITER_DBWD_LATE=4"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab.txt 2>&1; cat $O/ab.txt
DBM_LIB=$M DBM_ITER_DBWD_LATE=3 python3 tools/phases.py fused 2>&1 | grep -E "G:|D:" > $O/phases_late1.txt; cat $O/phases_late1.txt
