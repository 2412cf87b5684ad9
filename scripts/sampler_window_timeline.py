"""Where a SHORT timed window of the headline goes (the driver times 20 steps = 0.1 s; 1000 steps run 3-6 % faster per step): host timestamps of one
bench-shaped sample() call -- entry, each chain thread's start, its library call's enqueue interval, the join, the final synchronise -- next to the
GPU-side span of each chain (events on its stream).  usage (GPU box): python scripts/sampler_window_timeline.py [steps]"""
import sys, time, threading
import torch
sys.path.insert(0, ".")
sys.argv = [sys.argv[0]] + ["--steps", sys.argv[1] if len(sys.argv) > 1 else "20", "--warmup", "5", "--no-cpu-baseline", "--no-counters"]
import bench
from matinvent_amd import _lib
lib = _lib.load()
marks = []
real = lib.mi_sampler_run
def run(*a):
    t0 = time.perf_counter()
    rc = real(*a)
    marks.append((threading.current_thread().name, t0, time.perf_counter()))
    return rc
lib.mi_sampler_run = run
real_sample_one = None
from matinvent_amd import diffcsp
orig_one = diffcsp.DiffCSPModule._sample_one
ones = []
def one(self, *a, **k):
    t0 = time.perf_counter()
    r = orig_one(self, *a, **k)
    ones.append((threading.current_thread().name, t0, time.perf_counter()))
    return r
diffcsp.DiffCSPModule._sample_one = one
orig_sample = diffcsp.DiffCSPModule.sample
calls = []
def sample(self, *a, **k):
    marks.clear(); ones.clear()
    t0 = time.perf_counter()
    r = orig_sample(self, *a, **k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    calls.append((t0, t1, t2, list(marks), list(ones)))
    return r
diffcsp.DiffCSPModule.sample = sample
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
print(buf.getvalue()[:200])
for (t0, t1, t2, mk, on) in calls[-2:]:
    print(f"sample(): host returns after {(t1 - t0) * 1e3:.2f} ms, GPU idle after {(t2 - t0) * 1e3:.2f} ms")
    for (nm, a, b) in sorted(on, key=lambda r: r[1]):
        print(f"   {nm}: _sample_one from +{(a - t0) * 1e3:.2f} ms to +{(b - t0) * 1e3:.2f} ms")
    for (nm, a, b) in sorted(mk, key=lambda r: r[1]):
        print(f"   {nm}: mi_sampler_run enqueue from +{(a - t0) * 1e3:.2f} ms to +{(b - t0) * 1e3:.2f} ms")
