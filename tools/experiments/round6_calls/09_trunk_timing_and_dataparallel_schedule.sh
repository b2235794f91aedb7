#!/bin/bash
# round 6, call 9: (a) the per-wavefront cycle breakdown of the three persistent trunk kernels on round 6's code (-DTF_TIMING / -DTFB_TIMING
# builds of libdbm_measure.so, tools/experiments/chain_variants.sh) -- VERDICT r5 #2 asks for it next to every trunk experiment;
# (b) the data-parallel schedule on one GPU (one-rank RCCL communicator): tests + bench.py --force-comm against the default line
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c9; mkdir -p $O
bash tools/experiments/chain_variants.sh "chain_timing:-DTFB_TIMING" > $O/trunk_bwd_timing.txt 2>&1
SRC=trunk_fused bash tools/experiments/chain_variants.sh "fwd_timing:-DTF_TIMING" > $O/trunk_fwd_timing.txt 2>&1
tail -30 $O/trunk_bwd_timing.txt | cut -c1-220; tail -60 $O/trunk_fwd_timing.txt | cut -c1-260
timeout 1200 python3 -m pytest tests/test_gpu_parallel.py -x -q -m gpu > $O/tests_parallel.log 2>&1; tail -3 $O/tests_parallel.log
timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sweep --tables $O/t0.json > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sweep --force-comm --tables $O/t1.json > $O/bench_force_comm.json 2> $O/bench_force_comm.err; cut -c1-400 $O/bench_force_comm.json
