"""Per-shape A/B of igemm_conv_kernel's launch rules inside the training iteration.  Round 6: the switches it drives (DBM_IGEMM_OVERRIDE,
DBM_IGEMM_LOG) are TUNING switches and exist only in libdbm_measure.so (csrc/dbm_internal.h, DBM_TUNE_GETENV): build it first
(tools/build_measure.sh); the iterations are timed with tools/step_only.py (bench.py refuses any library but the product's)."""
import os, subprocess, sys, json, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEASURE = os.path.join(ROOT, "deepbedmap_amd", "libdbm_measure.so")
assert os.path.exists(MEASURE), "build deepbedmap_amd/libdbm_measure.so first: tools/build_measure.sh"
def bench(override, base_override=""):
    env = dict(os.environ, DBM_LIB=MEASURE)
    ov = ";".join(x for x in (base_override, override) if x)
    if ov: env["DBM_IGEMM_OVERRIDE"] = ov
    out = subprocess.run([sys.executable, "tools/step_only.py", "150"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
    return float(out.split()[1])
# distinct launches
env = dict(os.environ, DBM_IGEMM_LOG="1", DBM_LIB=MEASURE)
log = subprocess.run([sys.executable, "tools/step_only.py", "2"], cwd=ROOT, env=env,
                     capture_output=True, text=True, timeout=300).stderr
keys = {}
for m in re.finditer(r"igemm (\S+) tiles=(\d+) mt2_ok=(\d) -> mt2=(\d) waves=(\d+) ks=(\d+)", log):
    keys[m.group(1)] = dict(tiles=int(m.group(2)), ok=int(m.group(3)), mt2=int(m.group(4)), waves=int(m.group(5)), ks=int(m.group(6)))
base_ov = sys.argv[1] if len(sys.argv) > 1 else ""
cands = []
for k, c in sorted(keys.items()):
    cin = int(k.split(":")[1])
    alts = []
    if c["ok"]: alts.append((1 - c["mt2"], 4 if not c["mt2"] else -1, -1))
    if not c["mt2"]:
        for w in (4, 8, 16):
            if w != c["waves"] and cin % (2 * w) == 0 and c["ks"] == 1: alts.append((0, w, -1))
    if c["ks"] > 1:
        alts.append((c["mt2"], -1, c["ks"] * 2)); alts.append((c["mt2"], -1, max(1, c["ks"] // 2)))
    elif c["tiles"] <= 700 and cin >= 128:
        alts.append((c["mt2"], 4, 2))
    for a in alts: cands.append((k, c, a))
print(len(keys), "shapes,", len(cands), "candidates", flush=True)
b = [bench("", base_ov), bench("", base_ov)]
print("baseline", b, flush=True)
res = []
for i, (k, c, a) in enumerate(cands):
    ov = "%s=%d,%d,%d" % (k, a[0], a[1], a[2])
    t = [bench(ov, base_ov), bench(ov, base_ov)]
    if i % 8 == 7: b.append(bench("", base_ov))
    bm = sorted(b)[len(b) // 2]
    res.append((min(t) - bm, ov, c, t))
    print("%-40s cur=(%d,%d,%d) %.3f %.3f  delta_vs_base_median %.3f" % (ov, c["mt2"], c["waves"], c["ks"], t[0], t[1], min(t) - bm), flush=True)
print("baselines", [round(x, 3) for x in b])
for r in sorted(res)[:12]: print("BEST", r[1], round(r[0], 3), [round(x, 3) for x in r[3]])
