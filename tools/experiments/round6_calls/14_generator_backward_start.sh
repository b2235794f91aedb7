#!/bin/bash
# round 6, call 14: when does the generator's backward pass start?  Marks on chain[1]: retained forward done / loss terms / backward begin
# (behind the wait for the sampling lists, which are built on chain[0] behind the discriminator's fake-batch backward pass)
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c14; mkdir -p $O
python3 tools/phases.py fused > $O/phases.txt 2>&1; cat $O/phases.txt
DBM_ITER_CSR_EARLY=0 python3 tools/phases.py fused > $O/phases_csr_inside.txt 2>&1; grep -E "G:|joined" $O/phases_csr_inside.txt
