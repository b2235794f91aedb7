#!/bin/bash
# round 6, call 33: where the split-bf16 convolutions of the sweep's tail (conv_cl16x3_kernel: 815 us of a crop in six launches) spend
# their time (DBM_CL16X3_ABL, libdbm_measure.so, results wrong): 1 no MFMA loop, 2 no epilogue, 4 no staging after chunk 0
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c33; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for a in 0 1 2 4 3 5 6 7; do echo "abl $a:"; DBM_LIB=$M DBM_CL16X3_ABL=$a timeout 300 python3 tools/experiments/sweep_crop_ab.py x3_ 5 2>&1 | tail -6 | cut -c1-90; done > $O/abl.txt 2>&1; cat $O/abl.txt
