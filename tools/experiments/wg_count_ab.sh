#!/bin/bash
# workgroups per launch of the kernels that share the chip with the persistent trunk launches (they live on 64 CUs there)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
run() { echo "$* $(env "$@" timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"; }
run A=0
run DBM_WGRAD_DIRECT_WGS=256
run DBM_WGRAD_DIRECT_WGS=1024
run DBM_WGRAD_1X1_WGS=256
run DBM_WGRAD_SLOTS=256
run DBM_WGRAD_SLOTS=512
run DBM_WGRAD_SLOTS=768
run A=0
