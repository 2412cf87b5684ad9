"""A/B builds for timing diagnostics: recompile SOME translation units with extra flags and link them with the default objects into
matinvent_amd/lib/variants/libmatinvent_hip_<name>.so (git-ignored; selected at run time by MI_LIB_PATH, which matinvent_amd/_lib.py honours).
usage: python scripts/build_variant.py <name> <unit.hip>[,<unit.hip>...] <flag> [<flag> ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matinvent_amd import build as B  # noqa: E402

name, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build(verbose=False)   # the default objects
vdir = os.path.join(B.HERE, "lib", "variants")
os.makedirs(vdir, exist_ok=True)
objs = []
for src in B.SOURCES:
    obj = os.path.join(B.OBJ, src.replace(".hip", ".o"))
    if src in units:
        obj = os.path.join(vdir, f"{name}_{src.replace('.hip', '.o')}")
        subprocess.run(["/opt/rocm/bin/hipcc"] + B.ARCH + B.CFLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True, stderr=subprocess.DEVNULL)
    objs.append(obj)
out = os.path.join(vdir, f"libmatinvent_hip_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc"] + B.ARCH + ["-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", out], check=True)
print(out)
