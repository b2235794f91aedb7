#!/bin/bash
# round 6, call 29: the step loop of the LDS-window deformable kernel without its loads (DBM_X3W_ABL 110 = 14 + 32 + 64):
# + 128 no blend / split arithmetic, + 256 no MFMAs, both
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c29; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for a in 110 238 366 494 15; do echo "abl $a: $(DBM_LIB=$M DBM_X3W_ABL=$a timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 5 2>&1 | tail -1)"; done > $O/abl.txt 2>&1; cat $O/abl.txt
