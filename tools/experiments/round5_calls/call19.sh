#!/bin/bash
# round 5, call 19: helper workgroups re-measured on the current code (the discriminator's tail is now co-critical)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c19
bash tools/experiments/ab_env.sh "DBM_ITER_CSR_EARLY=0" "DBM_ITER_CSR_EARLY=0 DBM_TRUNK_HELPER=0" "DBM_ITER_CSR_EARLY=2" "DBM_ITER_CSR_EARLY=2 DBM_TRUNK_HELPER=0" "DBM_ITER_CSR_EARLY=0 DBM_ITER_EARLY_TWIN=1" > gpurun_out/r5c19/ab.txt 2>&1
cat gpurun_out/r5c19/ab.txt
