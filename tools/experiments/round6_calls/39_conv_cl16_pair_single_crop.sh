#!/bin/bash
# round 6, call 39: the two-workgroup form of the trunk layers for a SINGLE crop too (324 tiles of eight patches instead of 234 of eleven)?
# DBM_CL16_PAIR_MIN (libdbm_measure.so): smallest number of 16-row tiles that takes the form (default 4 x CUs)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c39; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for rep in 1 2; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_CL16_PAIR_MIN=1"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py "cl16_c64>32" 20 2>&1 | tail -2 | sed 's/.root.repo.deepbedmap_amd.//'; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
