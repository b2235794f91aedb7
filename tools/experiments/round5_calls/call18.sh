#!/bin/bash
# round 5, call 18: what the parts cost inside the iteration on the current code (libdbm_measure.so; results wrong by construction)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c18
L="DBM_LIB=$PWD/deepbedmap_amd/libdbm_measure.so DBM_ITER_CSR_EARLY=0"
bash tools/experiments/ab_env.sh "$L" "$L DBM_ITER_ABL=1" "$L DBM_ITER_ABL=2" "$L DBM_ITER_ABL=8" "$L DBM_ABL_SKIP=1" "$L DBM_ABL_NOPACK=1" "$L DBM_ABL_NOPACK=6" "$L DBM_ABL_SKIP=4" "$L DBM_ABL_SKIP=32" "$L DBM_ITER_ABL=4" "$L DBM_ITER_CSR_EARLY=2" > gpurun_out/r5c18/abl.txt 2>&1
cat gpurun_out/r5c18/abl.txt
