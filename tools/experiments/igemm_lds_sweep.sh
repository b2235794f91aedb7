#!/bin/bash
# NT x KC sweep of the LDS-staged igemm form on the 36x36 / 18x18 shapes (standalone durations)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for nt in 0 2 3 4 6 8; do for kc in 4 8; do
  echo "== NT=$nt KC=$kc"
  DBM_IGEMM_LDS_NT=$nt DBM_IGEMM_LDS_KC=$kc timeout 120 python tools/experiments/igemm_scaling.py 2>&1 | grep -E "36x36 Cout  64 Cin  (64|256)|18x18 Cout 128 Cin  64|18x18 Cout  64 Cin 128| 9x9  Cout 128 Cin 128"
done; done
echo "== old kernel"; DBM_IGEMM_LDS=0 timeout 120 python tools/experiments/igemm_scaling.py 2>&1 | grep -E "36x36 Cout  64 Cin  (64|256)|18x18 Cout 128 Cin  64|18x18 Cout  64 Cin 128| 9x9  Cout 128 Cin 128"
