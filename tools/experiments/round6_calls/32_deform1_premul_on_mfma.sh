#!/bin/bash
# round 6, call 32: the 64 -> 1 deformable layer's premultiplication (1x1 convolution 64 -> 9 tap planes) on v_mfma_f32_16x16x4f32
# instead of FMAs + cross-lane adds: parity tests, the sweep crop and the training step with both (DBM_DEFORM1_PREMUL_MFMA=0, measure lib)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c32; mkdir -p $O
M=$PWD/deepbedmap_amd/libdbm_measure.so
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
for rep in 1 2; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_DEFORM1_PREMUL_MFMA=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform1 10 2>&1 | tail -2 | sed 's/.root.repo.deepbedmap_amd.//'; done; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
for rep in 1 2; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_DEFORM1_PREMUL_MFMA=0"; do echo "[${e#*so}] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
