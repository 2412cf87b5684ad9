"""Build the gfx950 shared library in-tree:  python -m matinvent_amd.build

hipcc cross-compiles without a GPU; the resulting matinvent_amd/lib/libmatinvent_hip.so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmatinvent_hip.so")
SOURCES = ["cspnet.hip", "sampler.hip", "backward.hip", "graph.hip", "gemnet.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "matinvent_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile the library if it is missing or older than its sources.  Safe to call from several processes at once (one rank
    per GPU): an exclusive file lock serialises them, the compiler writes to a temporary file that is renamed into place, and a
    process that waited for the lock re-checks before compiling again."""
    if not force and not _stale():
        return LIB
    import fcntl
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            extra = os.environ.get("MI_EXTRA_FLAGS", "").split()  # tuning experiments (e.g. -DMI_UG=2 -DMI_RING=4)
            tmp = f"{LIB}.tmp{os.getpid()}"
            cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
