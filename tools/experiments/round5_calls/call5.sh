#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5c5; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv2d" > $O/pytest_ops.log 2>&1; tail -2 $O/pytest_ops.log
export DBM_LIB=$PWD/deepbedmap_amd/libdbm_measure.so
for A in 0 4; do
  DBM_CT_ABL=$A rocprofv3 --kernel-trace -f csv -d $O -o kt$A -- python3 tools/experiments/conv_tile_bench.py 3 > $O/kt$A.log 2>&1
  python3 - $O kt$A <<'PY'
import csv, sys
o, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f"{o}/{tag}_kernel_trace.csv")))
out = []
for r in rows:
    k = r["Kernel_Name"]
    if "conv_tile_kernel" in k:
        out.append("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(tag, " ".join(out[2::3]))
PY
done
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
