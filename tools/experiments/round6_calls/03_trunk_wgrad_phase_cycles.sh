#!/bin/bash
# round 6, call 3: where the trunk's weight-gradient launch spends its cycles (tools/wgrad_bench, -DDBM_WG_TIMING: staging / K loop / epilogue
# per workgroup) -- VERDICT r5 #3 ("stages 1.25 GB through LDS-DMA at half the fill rate")
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c3; mkdir -p $O
for n in 36 12; do timeout 120 tools/wgrad_bench/trunk $n; done > $O/trunk_wgrad_phases.txt 2>&1
timeout 120 tools/wgrad_bench/discriminator >> $O/trunk_wgrad_phases.txt 2>&1
cat $O/trunk_wgrad_phases.txt
