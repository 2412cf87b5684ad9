"""Build the gfx950 shared library in-tree:  python -m matinvent_amd.build

hipcc cross-compiles without a GPU; the resulting matinvent_amd/lib/libmatinvent_hip.so is
git-ignored but travels to the GPU box with the repo snapshot.  Every translation unit is compiled
to its own object file (in parallel, only when it or a header changed) and the objects are linked:
a cold build takes about as long as the slowest unit (~50 s) instead of the sum.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmatinvent_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
SOURCES = ["cspnet.hip", "node_chain.hip", "node_bwd.hip", "edge_stage.hip", "edge_fused.hip", "sampler.hip", "backward.hip", "graph.hip", "gemnet.hip"]
ARCH = ["--offload-arch=gfx950"]
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
# NO packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) in device code.  Measured on MI355X
# (scripts/force_fwd_repro.hip): a wave that executes them while waves of ANOTHER kernel with LDS traffic and
# MFMAs share its SIMD -- i.e. whenever two streams run concurrently, which the samplers' concurrent chains do -- intermittently gets
# wrong results in lanes 48-63 of a packed result.  The compiler forms these instructions by itself (SLP vectorisation of fp32 pairs);
# without the target feature it selects the scalar forms.  Cost: none measurable (50.0 / 50.1 against 49.4-50.4 structures/s on the
# headline, same box); effect: 120 of 120 trials of four concurrent MatterGen-shaped forwards bit-identical, against 103 of 120.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
if not os.environ.get("MI_ALLOW_PACKED_FP32"):   # (ablation only: rebuilds the library of rounds 1-3, e.g. to watch the concurrency tests fail)
    CFLAGS += NO_PACKED_FP32


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "matinvent_hip.h"), os.path.join(HERE, "..", "include", "matinvent_hip_debug.h"), __file__]


# Library variants: "" = the product library; "tf32" = the TF32-CLASS build (every plane-set product keeps its leading fp16 x fp16 term only:
# csrc/gemm_split.h MI_TF32_CLASS) behind bench.py's labelled `extra.tf32_class_path` line and tests/test_gpu_tf32_class.py -- never loaded by default.
VARIANTS = {"": [], "tf32": ["-DMI_TF32_CLASS=1"]}


def lib_path(variant: str = "") -> str:
    return LIB if not variant else os.path.join(HERE, "lib", f"libmatinvent_hip_{variant}.so")


def _flags_tag(variant: str = "") -> str:
    return " ".join(ARCH + CFLAGS + VARIANTS[variant] + os.environ.get("MI_EXTRA_FLAGS", "").split())


def _stale(variant: str = "") -> bool:
    """Missing, older than a source, or LINKED UNDER OTHER FLAGS than the ones in force now (the tag written next to the library): a
    library built with MI_ALLOW_PACKED_FP32=1 or MI_EXTRA_FLAGS must not keep being loaded once the variable is gone."""
    lib = lib_path(variant)
    if not os.path.exists(lib) or not os.path.exists(lib + ".flags"):
        return True
    with open(lib + ".flags") as f:
        if f.read() != _flags_tag(variant):
            return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "matinvent_hip.h"), os.path.join(HERE, "..", "include", "matinvent_hip_debug.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def _obj_stale(src: str, obj: str, flags_tag: str) -> bool:
    if not os.path.exists(obj) or not os.path.exists(obj + ".flags"):
        return True
    with open(obj + ".flags") as f:
        if f.read() != flags_tag:
            return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + _headers())


def build(force: bool = False, verbose: bool = True, variant: str = "") -> str:
    """Compile the library if it is missing or older than its sources.  Safe to call from several processes at once (one rank
    per GPU): an exclusive file lock serialises them, the linker writes to a temporary file that is renamed into place, and a
    process that waited for the lock re-checks before compiling again."""
    lib = lib_path(variant)
    if not force and not _stale(variant):
        return lib
    import fcntl
    objdir = OBJ if not variant else OBJ + "_" + variant
    os.makedirs(objdir, exist_ok=True)
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(variant):
                return lib
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            extra = VARIANTS[variant] + os.environ.get("MI_EXTRA_FLAGS", "").split()  # tuning experiments (e.g. -DMI_UG=2 -DMI_RING=4)
            tag = _flags_tag(variant)

            def compile_one(name: str) -> str:
                src, obj = os.path.join(CSRC, name), os.path.join(objdir, name.replace(".hip", ".o"))
                if force or _obj_stale(src, obj, tag):
                    cmd = [hipcc] + ARCH + CFLAGS + extra + ["-c", src, "-o", obj]
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
                    # (the host half of the compilation does not know the device-only target feature and says so: not a diagnostic of ours)
                    err = "\n".join(ln for ln in (r.stderr or "").splitlines() if "is not a recognized feature for this target" not in ln)
                    if err.strip():
                        print(err, file=sys.stderr, flush=True)
                    if r.returncode != 0:
                        raise subprocess.CalledProcessError(r.returncode, cmd)
                    with open(obj + ".flags", "w") as f:
                        f.write(tag)
                return obj

            with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(compile_one, SOURCES))
            tmp = f"{lib}.tmp{os.getpid()}"
            cmd = [hipcc] + ARCH + ["-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, lib)
                with open(lib + ".flags", "w") as f:
                    f.write(tag)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


if __name__ == "__main__":
    variants = [""] + (["tf32"] if "--tf32" in sys.argv or "--all" in sys.argv else [])
    for v in variants:
        print(build(force="--force" in sys.argv, variant=v))
