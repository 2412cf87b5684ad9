"""Fixed cost of one sampler call at the headline batch: timed K-step chains for several K (same process, warm caches)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
m = bench.build_module(dev)
class C: num_atoms = torch.tensor([bench.NATOM] * bench.B)
cb = C()
kw = dict(step_lr=bench.STEP_LR, streams=4)
m.sample(cb, seed=2, t_start=1000, t_stop=997, **kw)
final, _ = m.sample(cb, seed=1, t_start=1000, t_stop=1000, **kw)
state = (final["frac_coords"], final["lattices"], final["atom_types"])
for rep in range(2):
    for K in (0, 1, 2, 5, 10, 20, 40, 80, 160):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.sample(cb, seed=1, init=state, t_start=1000, t_stop=1000 - K, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"K={K:4d}: {dt * 1e3:8.2f} ms total, {dt * 1e3 / max(K, 1):7.3f} ms per step", flush=True)
