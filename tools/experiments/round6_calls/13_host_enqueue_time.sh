#!/bin/bash
# round 6, call 13: is the HOST the bottleneck?  (the profiled timeline shows the generator's stream idle for 0.5 ms between its forward and
# its loss kernel: the host was still enqueuing the discriminator's backward passes)  host_enqueue_time.py: enqueue time per iteration
# against device time per iteration, n = 4 / 16 / 64 iterations without a synchronisation
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c13; mkdir -p $O
for i in 1 2; do python3 tools/experiments/host_enqueue_time.py; done > $O/host_enqueue.txt 2>&1; cat $O/host_enqueue.txt
nproc; lscpu | grep -E "Model name|MHz" | head -3
