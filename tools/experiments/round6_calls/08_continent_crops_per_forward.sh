#!/bin/bash
# round 6, call 8: the full-continent bf16 sweep at 8 / 16 / 32 crops per forward (fuller launches: 180 trunk launches per forward whatever
# the crop count) -- does the fixed cost per launch amortise further?
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r6c8; mkdir -p $O
timeout 900 python3 tools/continent_sweep.py 8 16 32 24 > $O/continent.json 2> $O/continent.err; tail -3 $O/continent.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6c8/continent.json"))
for r in d["runs"]: print(r["dtype"], r["crops_per_batch"], "%.3f s" % r["sweep_s"], "%.3f ms/tile" % r["ms_per_tile"], r.get("max_diff_to_first_bf16_run_m"))
PY
