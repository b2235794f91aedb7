#!/bin/bash
# round 6, call 27: what the step loop of the LDS-window deformable kernel waits for (DBM_X3W_ABL with 14 = loop only:
# + 16 every step's weights are step 0's (L1-resident), + 32 no weight loads, + 64 no corner reads)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c27; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for a in 0 14 30 46 78 110 16 32; do echo "abl $a: $(DBM_LIB=$M DBM_X3W_ABL=$a timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 5 2>&1 | tail -1)"; done > $O/abl.txt 2>&1; cat $O/abl.txt
