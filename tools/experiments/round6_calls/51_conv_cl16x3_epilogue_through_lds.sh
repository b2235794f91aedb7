#!/bin/bash
# round 6, call 51: the split-bf16 convolutions' channels-last fp32 output through LDS (whole pixels per store): tests, crop fixtures,
# per-shape times and crop / continent numbers (no switch: compare with call 50's on the same kind of box)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c51; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_cl16.py -x -q -m gpu > $O/tests_cl16.log 2>&1; tail -2 $O/tests_cl16.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or continent" > $O/tests_config5.log 2>&1; tail -2 $O/tests_config5.log
for rep in 1 2; do timeout 300 python3 tools/experiments/sweep_crop_ab.py "x3_" 10 2>&1 | tail -6; done > $O/crop.txt 2>&1; cat $O/crop.txt
echo "$(timeout 600 python3 tools/continent_sweep.py 8 2>&1 | tail -1 | grep -o '"sweep_s": [0-9.]*, "ms_per_tile": [0-9.]*' | head -1)" > $O/continent.txt; cat $O/continent.txt
