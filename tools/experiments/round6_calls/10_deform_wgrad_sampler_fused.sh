#!/bin/bash
# round 6, call 10: final_conv_layer1's weight gradient with the sampler fused in (deform_wgrad64_fused_kernel: no 191 MB sample matrix
# written by the retained forward and read back by a 1x1 GEMM): parity (op-level deformable backward, model / full-size / DEM-range
# fixtures, the switch test), per-shape times, A/B inside the iteration (DBM_DEFORM_WGRAD_FUSED=0 = the sample matrix)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c10; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 2400 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_dem.py tests/test_gpu_round5.py -x -q -m gpu > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
for rep in 1 2 3; do for e in "DBM_X=1" "DBM_DEFORM_WGRAD_FUSED=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_deform_wgrad.txt 2>&1; cat $O/ab_deform_wgrad.txt
for e in "DBM_X=1" "DBM_DEFORM_WGRAD_FUSED=0"; do env $e timeout 300 python3 tools/experiments/step_shapes.py deform 10 2>&1 | grep -E "deform|wgrad_1x1|env" | sed "s/^/[$e] /"; done > $O/shapes.txt 2>&1; cat $O/shapes.txt
for e in "DBM_X=1" "DBM_DEFORM_WGRAD_FUSED=0"; do env $e timeout 300 python3 tools/experiments/step_shapes.py wgrad 10 2>&1 | grep -E "wgrad_1x1|deform_wgrad|env" | sed "s/^/[$e] /"; done >> $O/shapes.txt 2>&1; tail -6 $O/shapes.txt
