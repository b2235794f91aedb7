#!/bin/bash
# round 6, call 52: the continent at 8 / 12 / 16 / 24 crops per forward on the round's last code (12+: 2.8-4.5 s!) and which of the
# round's kernels is responsible (switches at 16 crops per forward)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c52; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0" "DBM_LIB=$M DBM_CL16_PAIR=0" "DBM_LIB=$M DBM_CL16X3_PAIR=0" "DBM_LIB=$M DBM_DEFORM1_PREMUL_MFMA=0"; do echo "[${e#*so }] $(env $e timeout 600 python3 tools/continent_sweep.py 16 2>&1 | tail -1 | grep -o '"sweep_s": [0-9.]*' | head -1)"; done > $O/which.txt 2>&1; cat $O/which.txt
