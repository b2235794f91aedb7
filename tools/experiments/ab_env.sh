#!/bin/bash
# usage: ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   -- tools/step_only.py 100 under each environment, twice, interleaved
# (tuning switches -- DBM_IGEMM_*, DBM_WGRAD_*, DBM_ITER_AUX, ... -- and the work-skipping ones exist only in libdbm_measure.so since round 6:
#  add DBM_LIB=$PWD/deepbedmap_amd/libdbm_measure.so to the environments that set them)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for rep in 1 2; do
  for e in "$@"; do
    echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done
