#!/bin/bash
# round 6, call 20: call 19's one-launch deep end with the layer table in DEVICE memory read through the constant address space
# (call 19's kernel indexed its 2.9 KB argument struct at run time -> the whole struct was copied to scratch, every descriptor field
# came from private memory) and two resident workgroups per CU (242 registers, no AGPR split).
# Quick verdict first (bitwise switch test, A/B), the wide suites only if it wins.
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c20; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "DISC_DEEP or batch_64" > $O/tests_switch.log 2>&1; tail -2 $O/tests_switch.log
for rep in 1 2 3; do for e in "DBM_X=1" "DBM_DISC_DEEP_FUSED=0"; do echo "[$e] $(env $e timeout 100 python3 tools/step_only.py 100 2>&1 | tail -1)"; done; done > $O/ab.txt 2>&1; cat $O/ab.txt
for e in "DBM_X=1" "DBM_DISC_DEEP_FUSED=0"; do env $e timeout 200 python3 tools/experiments/step_shapes.py "" 10 2>&1 | grep -E "disc_deep|_pm|c128>256|c256>256|c256>512|c512>512|env" | sed "s/^/[$e] /"; done > $O/shapes.txt 2>&1; cat $O/shapes.txt
