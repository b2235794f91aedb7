#!/bin/bash
# round 6, call 43: the whole -m gpu suite and __graft_entry__.smoke() on the round's last commit
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c43; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "pytest rc $?" >> $O/full_tests.log; tail -4 $O/full_tests.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
