"""Where a 20-step sample() call spends its wall time beside the steady state: host stamps around the call's phases and HIP events
on every chain's stream.  usage (GPU box): python scripts/window_breakdown.py [steps] [streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from matinvent_amd import _lib, build as _build, diffcsp

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
_build.build(verbose=False)
lib = _lib.load()
from matinvent_amd.cspnet import set_gemm_mode
set_gemm_mode("split")
dev = torch.device("cuda", 0)
m = bench.build_module(dev)
m.decoder.set_edge_mode("gemm")
B, NATOM, T = bench.B, bench.NATOM, bench.T

class Counts:
    num_atoms = torch.tensor([NATOM] * B)
cb = Counts()
skw = dict(step_lr=bench.STEP_LR, streams=S)
m.sample(cb, seed=1, t_start=T, t_stop=T - 5, **skw)
final, _ = m.sample(cb, seed=2, t_start=T, t_stop=T, **skw)
state = (final["frac_coords"], final["lattices"], final["atom_types"])
torch.cuda.synchronize()

stamps = {}
orig = diffcsp.DiffCSPModule._sample_one
def timed_one(self, *a, **k):
    import threading
    name = threading.current_thread().name
    st = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(st)
    h0 = time.perf_counter()
    r = orig(self, *a, **k)
    h1 = time.perf_counter()
    e1 = torch.cuda.Event(enable_timing=True); e1.record(st)
    stamps[name] = (h0, h1, e0, e1)
    return r
diffcsp.DiffCSPModule._sample_one = timed_one
import ctypes as C
PROF = int(os.environ.get("PROF", "0"))
for rep in range(int(os.environ.get("REPS", "3"))):
    stamps.clear()
    if PROF: _lib.check(lib.mi_profile_enable(m.decoder._h, 1))
    torch.cuda.synchronize()
    eb = torch.cuda.Event(enable_timing=True); eb.record()
    t0 = time.perf_counter()
    final, _ = m.sample(cb, seed=2, init=state, t_start=T, t_stop=T - K, **skw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if PROF:
        n_launch, tot_ms, union_ms = C.c_int64(), C.c_double(), C.c_double()
        _lib.check(lib.mi_profile_read(m.decoder._h, C.byref(n_launch), C.byref(tot_ms), C.byref(union_ms)))
        _lib.check(lib.mi_profile_enable(m.decoder._h, 0))
    print(f"rep {rep}: sample() returned after {1e3 * (t1 - t0):.2f} ms, synchronized after {1e3 * (t2 - t0):.2f} ms ({K} steps: {1e3 * (t2 - t0) / K:.3f} ms/step)")
    for name, (h0, h1, e0, e1) in sorted(stamps.items()):
        print(f"   {name}: host enter +{1e3 * (h0 - t0):.2f} ms, enqueue done +{1e3 * (h1 - t0):.2f} ms | GPU: chain start +{eb.elapsed_time(e0):.2f} ms, chain end +{eb.elapsed_time(e1):.2f} ms")
