"""node_chain.hip's L2 warm-up loads are fire-and-forget inline asm (`global_load_dword vD, ...` whose result nobody reads).  They are safe
only if EVERY such load of a kernel writes the ONE VGPR that stays reserved until the closing `s_waitcnt vmcnt(0)` ("+v"(touch_dummy) on
every asm, round 6; advisor finding of round 5).  This script compiles node_chain.hip to device assembly and checks, per kernel, that all
inline-asm dword loads target one register and that no other instruction between the first touch and the kernel's end writes it.
usage: python scripts/scan_touch_regs.py   (exit code 1 on a violation; tests/test_abi.py runs it)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matinvent_amd.build import ARCH, CSRC, NO_PACKED_FP32  # noqa: E402


def scan(extra=()):
    out = os.path.join(tempfile.mkdtemp(), "node_chain.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + ARCH + ["-O3", "-std=c++17", "-fno-gpu-rdc"] + NO_PACKED_FP32 + ["--cuda-device-only", "-S", os.path.join(CSRC, "node_chain.hip"), "-o", out]
                   + list(extra), check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    report, bad = [], []
    # kernels: "<name>:" label ... ".end_amdhsa_kernel"/s_endpgm; inline asm is bracketed by ";;#ASMSTART" / ";;#ASMEND"
    for m in re.finditer(r"^(_Z\w*node_chain\w*):[^\n]*\n(.*?)^\s+\.amdhsa_kernel ", text, flags=re.M | re.S):
        name, body = m.group(1), m.group(2)
        touches = re.findall(r";;#ASMSTART\s*\n\s*global_load_dword (v\d+), [^\n]*\n\s*;;#ASMEND", body)
        if not touches:
            continue
        regs = sorted(set(touches))
        first = body.find(";;#ASMSTART")
        last_wait = body.rfind("s_waitcnt vmcnt(0)")
        # any compiler-generated instruction in between that writes the register as a destination
        window = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body[first:last_wait], flags=re.S)
        clobbers = [ln.strip() for ln in window.splitlines() if re.match(r"\s+(v_|ds_read|ds_load|global_load|buffer_load|flat_load)\w*\s+" + re.escape(regs[0]) + r"\b", ln)
                    and not re.match(r"\s+v_mov_b32_e32 " + re.escape(regs[0]) + r", 0\s*$", ln)]   # (`touch_dummy = 0.f` on the paths that reach the touches: before them) if len(regs) == 1 else []
        report.append((name[:70], len(touches), regs, len(clobbers)))
        if len(regs) != 1 or clobbers:
            bad.append((name, regs, clobbers[:3]))
    return report, bad


if __name__ == "__main__":
    report, bad = scan(sys.argv[1:])
    for name, n, regs, c in report:
        print(f"{name}: {n} touch loads -> {','.join(regs)}; {c} other writes of that register before the closing wait")
    if not report:
        print("no touch loads found (pattern out of date?)")
        sys.exit(1)
    sys.exit(1 if bad else 0)
