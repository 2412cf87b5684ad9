"""Reference-default ragged batch (192 mp_20-sized crystals): structures/s against the number of concurrent chains."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from matinvent_amd.sampling import ATOM_DIST
dev = torch.device("cuda", 0)
m = bench.build_module(dev)
np.random.seed(0)
p = ATOM_DIST["mp_20"]
na = np.random.choice(len(p), 192, p=p)
class C: num_atoms = torch.tensor(na)
cb = C()
K = 40
for rep in range(2):
    for streams in (1, 2, 3, 4):
        m.sample(cb, seed=2, step_lr=5e-6, t_start=1000, t_stop=995, streams=streams)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.sample(cb, seed=1, step_lr=5e-6, t_start=1000, t_stop=1000 - K, streams=streams)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"chains {streams}: {dt * 1e3 / K:.3f} ms per step, {192 * K / (1000 * dt):.1f} structures/s", flush=True)
