"""Chains on the reference's default sampling batch (192 crystals with mp_20 atom counts): structures/s by the number of concurrent chains.
usage (GPU box): python scripts/default_batch_streams.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
np.random.seed(0)
from matinvent_amd.sampling import ATOM_DIST  # noqa: E402
p = ATOM_DIST["mp_20"]
na = np.random.choice(len(p), 192, p=p)
m = bench.build_module(dev)


class C:
    num_atoms = torch.tensor(na)


cb = C()
print("edges:", int((na.astype(np.int64) ** 2).sum()))
for rep in range(2):
    for st in (1, 2, 3, 4, 6):
        m.sample(cb, seed=2, step_lr=bench.STEP_LR, t_start=1000, t_stop=997, streams=st)
        final, _ = m.sample(cb, seed=1, step_lr=bench.STEP_LR, t_start=1000, t_stop=1000, streams=st)
        state = (final["frac_coords"], final["lattices"], final["atom_types"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.sample(cb, seed=1, step_lr=bench.STEP_LR, init=state, t_start=1000, t_stop=950, streams=st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"streams={st}: {192 * 50 / (1000 * dt):7.2f} structures/s", flush=True)
