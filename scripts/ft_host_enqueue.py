"""How long does the HOST take to enqueue one fine-tune micro-step (mi_ft_micro_step: ~300 launches)?  Wraps the library call with a wall
clock (no device synchronisation inside) and runs `bench.py --mode ft` with 1 / 2 / 4 concurrent groups.  If the enqueue time per call times
the number of groups approaches the step time, the multi-group line is bound by the launch path, not by the GPU.
usage (GPU box): python scripts/ft_host_enqueue.py"""
import json
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, threading, json
sys.path.insert(0, %r)
STUB, NG = sys.argv[2] == "1", int(sys.argv[1])
sys.argv = ["bench.py", "--mode", "ft", "--steps", "20", "--warmup", "3", "--ft-groups", sys.argv[1], "--no-cpu-baseline"]
from matinvent_amd import _lib
lib = _lib.load()
real = lib.mi_ft_micro_step
acc = {"n": 0, "t": 0.0, "gap": 0.0, "ngap": 0, "first": None, "last": None}
import atexit
def report():
    steps = max(1, (acc["n"] - 3 * NG) // NG - 1)
    print("HOST", json.dumps({"calls": acc["n"], "ms_in_the_call": 1e3 * acc["t"] / max(1, acc["n"]), "ms_of_python_between_calls": 1e3 * acc["gap"] / max(1, acc["ngap"]),
                              "ms_per_step_by_call_times": 1e3 * (acc["last"] - acc["first"]) / steps if acc["first"] else None}), flush=True)
atexit.register(report)
lock = threading.Lock()
last_exit = {}
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter()
        tid = threading.get_ident()
        r = 0 if (STUB and acc["n"] >= 3 * NG) else real(*a)   # (STUB: the timed region's calls return at once -> the step time is the host loop alone)
        t1 = time.perf_counter()
        with lock:
            if acc["n"] == 3 * NG:
                acc["first"] = t0
            if acc["n"] >= 3 * NG and (acc["n"] - 3 * NG) %% NG == 0:
                acc["last"] = t0
            acc["n"] += 1
            acc["t"] += t1 - t0
            if tid in last_exit and t0 - last_exit[tid] < 0.1:   # (python time between two micro-step calls of one thread)
                acc["gap"] += t0 - last_exit[tid]
                acc["ngap"] += 1
            last_exit[tid] = t1
        return r
lib.mi_ft_micro_step = Wrap()
import bench
try:
    bench.main()
except ZeroDivisionError:
    pass

''' % ROOT
for g, stub in (("1", "0"), ("4", "0"), ("1", "1"), ("4", "1")):
    r = subprocess.run([sys.executable, "-c", CHILD, g, stub], capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    host = [ln for ln in r.stdout.splitlines() if ln.startswith("HOST")]
    d = json.loads(line[-1]) if line else {}
    print(("HOST LOOP ALONE (library call stubbed) " if stub == "1" else "") + f"groups={g}: {d.get('value', 0):.0f} crystal-timesteps/s, {d.get('ms_per_step', 0):.2f} ms per step;", host[-1] if host else r.stderr[-300:])
