#!/bin/bash
# round 6, call 18: 1500 fused iterations twice from one seed (bitwise: a race in the round's new kernels / stream edges would show) and
# trainer's end-to-end epoch throughput (tools/epoch_bench.py)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c18; mkdir -p $O
timeout 900 python3 tools/experiments/soak_determinism.py 1500 > $O/soak_determinism.txt 2>&1; tail -4 $O/soak_determinism.txt | cut -c1-300
timeout 900 python3 tools/epoch_bench.py 3 > $O/epoch_bench.txt 2>&1; tail -6 $O/epoch_bench.txt
