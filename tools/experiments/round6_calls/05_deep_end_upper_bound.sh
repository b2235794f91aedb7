#!/bin/bash
# round 6, call 5: what the discriminator's deep end (conv_layer5..9 + their BatchNorm launches + the linear layers) costs INSIDE the
# iteration -- libdbm_measure.so, DBM_D_ABL (results wrong): the upper bound of what one fused launch per pass can bring (VERDICT r5 #1a);
# and the LDS-DMA landing layout of global_load_lds_dwordx3 (VERDICT r5 #8b)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c5; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
bash tools/experiments/ab_env.sh "DBM_LIB=$M" "DBM_LIB=$M DBM_D_ABL=1" "DBM_LIB=$M DBM_D_ABL=2" "DBM_LIB=$M DBM_D_ABL=8" "DBM_LIB=$M DBM_D_ABL=3" "DBM_LIB=$M DBM_D_ABL=11" "DBM_LIB=$M DBM_ITER_ABL=1" > $O/ab_deep_end.txt 2>&1; cat $O/ab_deep_end.txt
tools/experiments/ubench/lds_dma_b96 > $O/lds_dma_b96.txt 2>&1; cat $O/lds_dma_b96.txt
