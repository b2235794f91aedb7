#!/bin/bash
# round 6, call 4: the trunk's weight gradients double buffered (wgrad_wave_dma2_kernel: one image per band, the next image's LDS-DMA in
# flight under the K loop, one wavefront per SIMD): parity (op-level conv backward, fused trunk fwd/bwd, the batch-64 fixtures), the
# stand-alone phase cycles again, and the A/B inside the iteration (DBM_WGRAD_DMA2=0 in libdbm_measure.so = the single-buffered kernel)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c4; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv2d" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 1800 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_dem.py -x -q -m gpu > $O/tests_model.log 2>&1; tail -3 $O/tests_model.log
for n in 36 12; do timeout 120 tools/wgrad_bench/trunk $n; done > $O/trunk_wgrad_phases.txt 2>&1; cat $O/trunk_wgrad_phases.txt
M=$PWD/deepbedmap_amd/libdbm_measure.so
for rep in 1 2 3; do for e in "DBM_LIB=$M" "DBM_LIB=$M DBM_WGRAD_DMA2=0"; do echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab_dma2.txt 2>&1; cat $O/ab_dma2.txt
for e in "DBM_X=1" "DBM_WGRAD_DMA2=0"; do env DBM_LIB=$M $e timeout 300 python3 tools/experiments/step_shapes.py wave_dma 10 2>&1 | grep -E "wave_dma|env" | sed "s/^/[$e] /"; done > $O/shapes_dma2.txt 2>&1; cat $O/shapes_dma2.txt
