#!/bin/bash
# usage: ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   -- tools/step_only.py 100 under each environment, twice, interleaved
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for rep in 1 2; do
  for e in "$@"; do
    echo "[$e] $(env $e timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done
