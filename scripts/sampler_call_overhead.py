"""Fixed cost of one DiffCSPModule.sample call at the headline shape (four concurrent chains): wall time of k-step calls (k = 1, 2, 5, 20, 40),
the host time until sample() returns (no device wait), and a per-phase host clock of one chain's thread.  usage (GPU box): python scripts/sampler_call_overhead.py"""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np, torch
import bench

dev = torch.device("cuda", 0)
m = bench.build_module(dev)
B, n, T = bench.B, bench.NATOM, bench.T
class Counts:
    num_atoms = torch.tensor([n] * B)
cb = Counts()
kw = dict(step_lr=bench.STEP_LR, streams=4)
m.sample(cb, seed=1, t_start=T, t_stop=T - 3, **kw)
final, _ = m.sample(cb, seed=2, t_start=T, t_stop=T, **kw)
state = (final["frac_coords"], final["lattices"], final["atom_types"])
torch.cuda.synchronize()
res = {}
for k in (1, 2, 5, 20, 40):
    ts, hs = [], []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.sample(cb, seed=3, init=state, t_start=T, t_stop=T - k, **kw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(t2 - t0); hs.append(t1 - t0)
    res[k] = (1e3 * min(ts), 1e3 * min(hs))
    print(f"{k:3d} steps: {res[k][0]:8.2f} ms wall (best of 5), host returns after {res[k][1]:7.2f} ms")
slope = (res[40][0] - res[20][0]) / 20
print(f"steady state {slope:.3f} ms per step; fixed cost of a call = {res[20][0] - 20 * slope:.2f} ms (from the 20-step call), {res[1][0] - slope:.2f} ms (from the 1-step call)")
