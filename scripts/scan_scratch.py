"""Every kernel of the default library with its scratch size and spill counts (none should have any), and the number of packed-fp32
VALU instructions in its device code (none either: matinvent_amd/build.py says why): compiles each translation unit to
device assembly (hipcc cross-compiles without a GPU) and reads the amdhsa.kernels metadata.  usage: python scripts/scan_scratch.py [extra flags]"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matinvent_amd.build import ARCH, CSRC, NO_PACKED_FP32, SOURCES  # noqa: E402

extra = sys.argv[1:]
tmp = tempfile.mkdtemp()


def asm(name):
    out = os.path.join(tmp, name.replace(".hip", ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc"] + ARCH + ["-O3", "-std=c++17", "-fno-gpu-rdc"] + NO_PACKED_FP32 + ["--cuda-device-only", "-S", os.path.join(CSRC, name), "-o", out] + extra,
                   check=True, stderr=subprocess.DEVNULL)
    return out


with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
    files = list(pool.map(asm, SOURCES))
total, bad, packed = 0, [], 0
for f in files:
    s = open(f).read()
    packed += len(re.findall(r"^\s+v_pk_(?:fma|mul|add)_f32|^\s+v_pk_mov_b32", s, flags=re.M))   # (must be none: see matinvent_amd/build.py)
    meta = s[s.find("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count")[1:]:
        get = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        total += 1
        if int(get("private_segment_fixed_size")) or int(get("vgpr_spill_count")):
            bad.append((os.path.basename(f), get("name"), get("private_segment_fixed_size"), get("vgpr_spill_count"), get("vgpr_count")))
print(f"{total} kernels, {len(bad)} with scratch or spilled vector registers; {packed} packed-fp32 instructions")
for b in bad:
    print("  %s  %s  scratch %s B, %s VGPR spills, %s VGPRs" % b)
sys.exit(1 if bad or packed else 0)
