#!/bin/bash
# round 6, call 41: the driver's multi-GPU launch line with ONE rank (torch.distributed.run) and the one-rank RCCL schedule (--force-comm)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c41; mkdir -p $O
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-sweep --no-continent > $O/torchrun_n1.json 2> $O/torchrun_n1.err; echo "rc $?"; cut -c1-400 $O/torchrun_n1.json; tail -3 $O/torchrun_n1.err
timeout 900 python3 bench.py --steps 100 --warmup 10 --force-comm --no-cpu-baseline --no-sweep --no-continent > $O/force_comm.json 2> $O/force_comm.err; echo "rc $?"; cut -c1-300 $O/force_comm.json; tail -3 $O/force_comm.err
timeout 900 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sweep --no-continent > $O/plain.json 2> $O/plain.err; echo "rc $?"; cut -c1-300 $O/plain.json; tail -2 $O/plain.err
