#!/usr/bin/env python3
"""Static scan for store drains: `vmcnt` counts loads AND stores, in order, so a load issued behind a store can only be awaited by
draining the store (a write round trip, ~2 us).  For every kernel of every .hip source: linear instruction order, sequences
    <global / buffer store> ... <global / buffer load> ... s_waitcnt vmcnt(N)   with N smaller than the loads issued since that store
i.e. waits that cannot be satisfied without the store having completed.  Loop back-edges and branches are ignored (linear order), so the
count is a lower bound inside loops and an upper bound across exclusive branches: a pointer to read the ISA, not a measurement.
(Round 5: real and costly in conv_tile's first epilogue, conv_cl16, the fused deformable kernels and the training-mode BatchNorm
kernels, whose guarded stores each sat behind a vmcnt(0): the branch-free forms with as many register slots as the plane needs are
25-30 % shorter.  A first branch-free BatchNorm rewrite that kept the general kernel's eight slots per image was SLOWER -- three
quarters of its buffer accesses were dummies on the 9 x 9 planes -- so: a pointer to read the ISA, then measure.)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deepbedmap_amd", "csrc")
STORE = re.compile(r"^\s*(global_store|buffer_store|flat_store|global_atomic|buffer_atomic|flat_atomic)")
LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load)")   # (LDS-DMA forms included: they count in vmcnt too)
WAIT = re.compile(r"s_waitcnt.*vmcnt\((\d+)\)")
FUNC = re.compile(r"^(_Z\w+):")


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "a.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", asm, path],
                           capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            print(f"{os.path.basename(path)}: compile failed\n{r.stderr[-400:]}")
            return
        func, out = None, {}
        pending = []   # outstanding vector-memory operations in issue order: 's' or 'l'
        for line in open(asm):
            m = FUNC.match(line)
            if m:
                func, pending = m.group(1), []
                continue
            if func is None:
                continue
            if "s_endpgm" in line:
                func = None
                continue
            if STORE.match(line):
                pending.append("s")
            elif LOAD.match(line):
                pending.append("l")
            else:
                w = WAIT.search(line)
                if w:
                    keep = int(w.group(1))
                    done, pending = (pending[:len(pending) - keep], pending[len(pending) - keep:]) if keep < len(pending) else ([], pending)
                    # a drain: the completed prefix holds a store that is FOLLOWED by a load inside the prefix (the wait was for that load)
                    if "s" in done and "l" in done[done.index("s"):]:
                        out[func] = out.get(func, 0) + 1
        name = subprocess.run(["c++filt"] + list(out), capture_output=True, text=True).stdout.split("\n") if out else []
        for (f, n), d in zip(out.items(), name):
            print(f"{os.path.basename(path):24s} {n:3d}  {d.replace('(anonymous namespace)::', '').split('(')[0][:110]}")
        if not out:
            print(f"{os.path.basename(path):24s}   0")


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        scan(f)
