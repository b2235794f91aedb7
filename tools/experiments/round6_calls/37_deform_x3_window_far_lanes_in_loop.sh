#!/bin/bash
# round 6, call 37: the LDS-window deformable kernel serving far samples (offsets beyond the window) inside the pipelined loop -- per lane,
# from global memory under the complementary EXEC mask -- instead of sending the whole wavefront to an unpipelined loop: bitwise tests,
# the crop with zero offsets (reference initialisation) and the continent with offsets of about a pixel (tools/dem_model.py)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c37; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "deform" > $O/tests_ops.log 2>&1; tail -2 $O/tests_ops.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or continent" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do env $e timeout 300 python3 tools/experiments/sweep_crop_ab.py deform64 10 2>&1 | tail -2; done > $O/ab_crop.txt 2>&1; cat $O/ab_crop.txt
for e in "DBM_X=1" "DBM_DEFORM_X3_WINDOW=0"; do echo "[$e] $(env $e timeout 600 python3 tools/continent_sweep.py 8 2>&1 | tail -1 | cut -c1-300)"; done > $O/ab_continent.txt 2>&1; cat $O/ab_continent.txt
