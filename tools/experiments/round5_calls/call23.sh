#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c23
rm -f gpurun_out/r5c23/out.txt
timeout 600 python3 -m pytest "tests/test_gpu_model.py::test_discriminator_forward_backward_parity" -x -q 2>&1 | grep -E "AssertionError|passed|failed" | head -12 >> gpurun_out/r5c23/out.txt
cat gpurun_out/r5c23/out.txt
