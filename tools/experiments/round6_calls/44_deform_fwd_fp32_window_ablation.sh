#!/bin/bash
# round 6, call 44: where deform_conv64_fusedw_kernel's 119 us go (DBM_FUSEDW_ABL, libdbm_measure.so: 1 no step loop, 2 no staging,
# 8 no stores, 256 no MFMAs)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c44; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
for a in 0 1 2 8 256 11 10; do echo "abl $a: $(DBM_LIB=$M DBM_DEFORM_FWD_WINDOW=1 DBM_FUSEDW_ABL=$a timeout 300 python3 tools/experiments/step_shapes.py deform64 5 2>&1 | grep deform64 | head -1)"; done > $O/abl.txt 2>&1; cat $O/abl.txt
