#!/bin/bash
# round 6, call 35: call 34's A/B again over 30 crops (the in-flow effect of the two-workgroup form, not the bracketed launches)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c35; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for rep in 1 2 3; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_CL16X3_PAIR=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py "x3_c64>18" 30 2>&1 | tail -2 | sed 's/.root.repo.deepbedmap_amd.//'; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
