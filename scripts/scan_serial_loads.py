"""Kernels whose ISA holds CHAINS of `global_load_dword* ... s_waitcnt vmcnt(0)` a few instructions apart -- the shape `cond ? table[i] : 0` compiles to
(one exec-masked branch per element with its own wait: N dependent memory latencies where one would do; round 5 found eight of them at the head of the
pair epilogue and four at the head of the node chain).  usage: python scripts/scan_serial_loads.py [min chain length]"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matinvent_amd.build import ARCH, CSRC, NO_PACKED_FP32, SOURCES  # noqa: E402

minlen = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tmp = tempfile.mkdtemp()


def asm(name):
    out = os.path.join(tmp, name.replace(".hip", ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc"] + ARCH + ["-O3", "-std=c++17", "-fno-gpu-rdc"] + NO_PACKED_FP32 + ["--cuda-device-only", "-S", os.path.join(CSRC, name), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return out


with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
    files = list(pool.map(asm, SOURCES))
for f in files:
    kern, ops = None, []
    for ln in open(f):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern, ops = m.group(1), []
            continue
        t = ln.strip().split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        if t.startswith("s_endpgm") and kern:
            # chains: a wait vmcnt(0) within 8 instructions after a (global|buffer)_load_dword(x2)? that itself follows the previous such wait closely
            chain, best, last_wait, pending = 0, 0, -100, -100
            for k, op in enumerate(ops):
                if re.match(r"(global|buffer)_load_dword(x2)?\b", op):
                    pending = k
                elif op.startswith("s_waitcnt") and "vmcnt(0)" in op and k - pending <= 8:
                    chain = chain + 1 if pending - last_wait <= 40 else 1
                    best = max(best, chain)
                    last_wait = k
                    pending = -100
            if best >= minlen:
                print(f"{os.path.basename(f):16s} {best:3d} dependent load+wait steps in a row   {kern[:100]}")
            kern = None
            continue
        ops.append(t)
