#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5c4; mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O -o kt -- python3 tools/experiments/conv_tile_bench.py 5 > $O/kt.log 2>&1

rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv -d $O -o pmc1 -- python3 tools/experiments/conv_tile_bench.py 2 > $O/pmc1.log 2>&1

python3 - $O <<'PY'
import csv, sys, collections
o = sys.argv[1]
for tag in ("kt",):
    rows = list(csv.DictReader(open(f"{o}/{tag}_kernel_trace.csv")))
    print(tag)
    for r in rows:
        k = r["Kernel_Name"]
        if "conv_tile_kernel" in k or "igemm_conv_kernel" in k:
            print("  %-70s grid %6s  %.1f us" % (k[:70], r.get("Grid_Size_X", r.get("Grid_Size")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for tag in ("pmc1",):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f"{o}/{tag}_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "conv_tile_kernel" not in k:
            continue
        key = (r["Dispatch_Id"], k[:60], r["Grid_Size"])
        acc.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print(tag)
    for key, v in acc.items():
        print("  ", key[1], key[2], {a: int(b) for a, b in v.items()})
PY
rm -f $O/*_kernel_trace.csv $O/*_counter_collection.csv $O/*agent_info.csv
