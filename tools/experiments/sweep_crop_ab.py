#!/usr/bin/env python3
"""bf16 (and fp32) time of one 288 x 288 sweep crop under the current environment, plus the per-shape rows whose tag contains a filter.

    python tools/experiments/sweep_crop_ab.py [filter] [crops]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flt = sys.argv[1] if len(sys.argv) > 1 else "deform"
env = dict(os.environ, CROPS=sys.argv[2] if len(sys.argv) > 2 else "10")
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_crop_bench.py")], env=env, capture_output=True, text=True, timeout=600)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    sys.exit(out.stdout[-1000:] + out.stderr[-2000:])
res = json.loads(line[-1])
sw = {k: v for k, v in os.environ.items() if k.startswith("DBM_")}
print(f"env {sw}: bf16 {res['bf16']['ms_per_crop']:.3f} ms per crop, fp32 {res['fp32']['ms_per_crop']:.3f}")
for r in res["bf16"].get("per_shape_standalone", []):
    if flt in r["shape"]:
        print(f"   {r['shape']:32s} x{r['launches']} {1e3 * r['ms'] / r['launches']:9.1f} us")
