"""Static ISA audit: per kernel, how many `s_waitcnt vmcnt(N)` wait for a vector load that was issued at most W instructions earlier (an EXPOSED memory
latency unless another wave covers it), and how many of them sit outside loops.  Finds what round 5 found by hand: compiler-serialised
`cond ? table[i] : 0` chains, waits forced by register copies, tables read behind a burst of operand loads.
usage: python scripts/isa_tight_waits.py [kernel-name-substring ...]"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matinvent_amd.build import ARCH, CSRC, NO_PACKED_FP32, SOURCES  # noqa: E402

want = sys.argv[1:]
W = 12
tmp = tempfile.mkdtemp()


def asm(name):
    out = os.path.join(tmp, name.replace(".hip", ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc"] + ARCH + ["-O3", "-std=c++17", "-fno-gpu-rdc"] + NO_PACKED_FP32 + ["--cuda-device-only", "-S", os.path.join(CSRC, name), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return out


with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
    files = list(pool.map(asm, SOURCES))
rows = []
for f in files:
    txt = open(f).read()
    # kernels: from "name:" to the .end_amdhsa_kernel / next function; use ".Lfunc_end" markers
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want and not any(w in name for w in want):
            continue
        ops = []
        for ln in body.split("\n"):
            t = ln.split(";")[0].strip()
            if t and not t.startswith("."):
                ops.append(t)
        issued, last_load, tight = 0, -1000, 0
        for k, op in enumerate(ops):
            if re.match(r"(global|buffer|flat)_load_", op) and " lds" not in op:
                issued += 1
                last_load = k
            elif op.startswith("s_waitcnt") and "vmcnt(" in op:
                n = int(re.search(r"vmcnt\((\d+)\)", op).group(1))
                if n == 0 and k - last_load <= W:
                    tight += 1
        if tight >= 2:
            rows.append((tight, os.path.basename(f), name))
for tight, f, name in sorted(rows, reverse=True)[:60]:
    print(f"{tight:4d} tight vmcnt(0) waits  {f:14s} {name[:110]}")
