#!/bin/bash
# Runs on the GPU box: HBM traffic of the dominant kernel from PMC counters, one counter per pass (MI355X_MICROARCH.md
# "HBM": FETCH_SIZE / WRITE_SIZE are in KiB, separate passes; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x).
# usage: tools/pmc_traffic.sh <tag>
set -e
TAG=${1:-pmc}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "${ROOT:?repository root not found}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d "$OUT" -o $C -- python3 bench.py --no-cpu-baseline --no-continent --tables "$OUT/tables.json" --steps 2 --warmup 1 > "$OUT/$C.log" 2>&1 || { echo "rocprofv3 failed; last lines of "$OUT/$C.log":" >&2; tail -n 30 "$OUT/$C.log" >&2; exit 1; }
done
# ("$OUT/tables.json": the per-shape tables of the last counter pass -- its own file, bench.py --tables)
python3 - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]


def family(kn):
    return ("igemm_conv_kernel" if ("igemm_conv_kernel" in kn or "igemm_pm_kernel" in kn or "conv_tile_kernel" in kn or "input_block_fused_kernel" in kn or "deform_conv64_fused" in kn or "deform_bwd64_fused" in kn)
            else "trunk_fused_bwd_kernel" if "trunk_fused_bwd_kernel" in kn
            else "trunk_fused_kernel_helper" if "trunk_fused_kernel<27, true>" in kn  # the form with a helper workgroup per image
            else "trunk_fused_kernel" if "trunk_fused_kernel" in kn and "pack" not in kn
            else "wgrad_kernel" if ("wgrad_" in kn or "deform_wgrad64_fused" in kn) else None)


def shape_class(kn):
    """the class a launch belongs to in bench.py's per-shape tables (roofline.per_shape / extras.sweep.*.per_shape_standalone)"""
    for pat, c in (("igemm_conv_kernel", "igemm"), ("igemm_pm_kernel", "igemm"), ("conv_tile_kernel", "igemm"), ("input_block_fused_kernel", "igemm"), ("deform_conv64_fused", "deform64"), ("deform_conv64_x3", "deform64x3"), ("deform_conv1_fused", "deform1"),
                   ("deform1_premul", "deform1"), ("deform1_sample", None),
                   ("deform_bwd64_fused", "deform_bwd64"), ("deform_wgrad64_fused", "wgrad"), ("deform_wgrad64_fold", None), ("conv_cl16x3_kernel", "x3"), ("conv_cl16_kernel", "cl16"),
                   ("trunk_fused_bwd_kernel", "trunk_bwd"), ("trunk_fused_kernel", "trunk_fwd"), ("wgrad_pair_fold", None), ("wgrad_fold", None),
                   ("wgrad_", "wgrad")):
        if pat in kn and "pack" not in kn:
            return c
    return None


def tag_class(tag):
    for pre, c in (("deform_bwd64", "deform_bwd64"), ("deform_wgrad64", "wgrad"), ("deform64x3", "deform64x3"), ("deform64", "deform64"), ("deform", "deform1"), ("x3_", "x3"), ("cl16_", "cl16"),
                   ("trunk_fwd", "trunk_fwd"), ("trunk_bwd", "trunk_bwd"), ("input_block", "igemm"), ("c", "igemm")):
        if tag.startswith(pre):
            return c
    return "wgrad"


# the launch shapes of the TRAINING step (bench.py's roofline.per_shape of the same command): the family averages below count
# these only -- the command's inference leg launches the same kernels on much larger grids
train_keys = None
try:
    train_keys = {(tag_class(r["shape"]), r["workgroups"]) for r in json.load(open(f"{out}/tables.json"))["per_shape"]}
except Exception:
    pass

res = {}
shapes = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {"FETCH_SIZE": 0, "WRITE_SIZE": 0}, "kernels": set()})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"{out}/{c}_counter_collection.csv")):
        if r["Counter_Name"] != c:
            continue
        kn = r["Kernel_Name"]
        k = family(kn)
        sc = shape_class(kn)
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        if k and (train_keys is None or (sc, wgs) in train_keys):
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        if sc:
            sh = shapes[(sc, wgs)]
            sh[c] += float(r["Counter_Value"]); sh["n"][c] += 1; sh["kernels"].add(kn.replace("void ", "").split("(")[0][:60])
    for k, (s, n) in acc.items():
        res.setdefault(k, {})[c + "_KiB_per_launch"] = s / n
        res[k]["launches_sampled"] = n
for k, v in res.items():
    # gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B -> x2 (upper bound for partially coalesced gathers)
    v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE_KiB_per_launch"] + v["WRITE_SIZE_KiB_per_launch"]) * 1024.0

# ---- per launch shape, against the ALGORITHMIC bytes bench.py's brackets carry (same command: its JSON line is in the log) ----
alg = collections.defaultdict(lambda: [0.0, 0, set()])  # (class, workgroups) -> [algorithmic bytes summed, launches, tags]
try:
    bench = json.load(open(f"{out}/tables.json"))
    tables = [bench.get("per_shape", [])]
    for leg in ("fp32", "bf16"):
        tables.append(((bench.get("sweep") or {}).get(leg) or {}).get("per_shape_standalone", []))
    for t in tables:
        for row in t:
            a = alg[(tag_class(row["shape"]), row["workgroups"])]
            a[0] += row["algorithmic_bytes_per_launch"] * row["launches"]; a[1] += row["launches"]; a[2].add(row["shape"])
except Exception as e:  # pragma: no cover
    print("no bench line to join:", repr(e), file=sys.stderr)
per_shape = []
for (sc, wgs), sh in shapes.items():
    if not sh["n"]["FETCH_SIZE"] or not sh["n"]["WRITE_SIZE"]:
        continue
    hbm = (2.0 * sh["FETCH_SIZE"] / sh["n"]["FETCH_SIZE"] + sh["WRITE_SIZE"] / sh["n"]["WRITE_SIZE"]) * 1024.0
    row = {"class": sc, "workgroups": wgs, "kernels": sorted(sh["kernels"]), "launches_sampled": sh["n"]["FETCH_SIZE"],
           "fetch_KiB_per_launch": sh["FETCH_SIZE"] / sh["n"]["FETCH_SIZE"], "write_KiB_per_launch": sh["WRITE_SIZE"] / sh["n"]["WRITE_SIZE"],
           "hbm_bytes_per_launch": hbm}
    a = alg.get((sc, wgs))
    if a and a[1]:
        row["shapes"] = sorted(a[2])
        row["algorithmic_bytes_per_launch"] = a[0] / a[1]
        row["traffic_over_algorithmic"] = hbm / max(a[0] / a[1], 1.0)
    per_shape.append(row)
per_shape.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches_sampled"])
# family totals against their algorithmic bytes (launch-weighted over the family's shapes)
fam_class = {"igemm_conv_kernel": ("igemm", "deform64", "deform_bwd64"), "wgrad_kernel": ("wgrad",), "trunk_fused_kernel": ("trunk_fwd",),
             "trunk_fused_kernel_helper": ("trunk_fwd",), "trunk_fused_bwd_kernel": ("trunk_bwd",)}
for k, v in res.items():
    rows = [r for r in per_shape if r["class"] in fam_class.get(k, ()) and "algorithmic_bytes_per_launch" in r and
            (train_keys is None or (r["class"], r["workgroups"]) in train_keys)]
    if k.startswith("trunk_fused_kernel"):
        rows = [r for r in rows if any(("helper" in s) == k.endswith("helper") for s in r["shapes"])]
    n = sum(r["launches_sampled"] for r in rows)
    if n:
        v["algorithmic_bytes_per_launch"] = sum(r["algorithmic_bytes_per_launch"] * r["launches_sampled"] for r in rows) / n
        v["traffic_over_algorithmic"] = sum(r["hbm_bytes_per_launch"] * r["launches_sampled"] for r in rows) / n / v["algorithmic_bytes_per_launch"]
res["per_shape"] = per_shape
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_shape"}, indent=1))
for r in per_shape[:40]:
    print(r["class"], r["workgroups"], r.get("shapes"), "hbm %.2f MB" % (r["hbm_bytes_per_launch"] / 1e6),
          "alg %.2f MB x%.2f" % (r.get("algorithmic_bytes_per_launch", 0) / 1e6, r.get("traffic_over_algorithmic", 0)))
PY
rm -f "$OUT"/*_kernel_trace.csv "$OUT"/*_counter_collection.csv
