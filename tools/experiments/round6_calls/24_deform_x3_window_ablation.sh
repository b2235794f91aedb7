#!/bin/bash
# round 6, call 24: where the LDS-window deformable kernel's 511 us go (DBM_X3W_ABL, libdbm_measure.so, results wrong):
# 1 no step loop, 2 no window staging, 4 no geometry, 8 no stores
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c24; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for a in 0 1 2 4 8 3 5 7 15 14; do DBM_LIB=$M DBM_X3W_ABL=$a timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 5 2>&1 | tail -2; done > $O/abl.txt 2>&1; cat $O/abl.txt
