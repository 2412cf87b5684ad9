"""Workload for scripts/window_steps.py: warm-up, synchronise, then TWO 20-step sample() calls back to back with no synchronisation between them (the GPU
never idles between the calls: is the slow start of a call the idle gap in front of it, or the chains starting at the same point of their step?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from matinvent_amd import _lib, build as _build
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
skw = dict(step_lr=bench.STEP_LR, streams=int(os.environ.get("STREAMS", "4")))
m.sample(cb, seed=1, t_start=T, t_stop=T - 5, **skw)
final, _ = m.sample(cb, seed=2, t_start=T, t_stop=T, **skw)
state = (final["frac_coords"], final["lattices"], final["atom_types"])
torch.cuda.synchronize()
f1, _ = m.sample(cb, seed=2, init=state, t_start=T, t_stop=T - 20, **skw)
f2, _ = m.sample(cb, seed=2, init=(f1["frac_coords"], f1["lattices"], f1["atom_types"]), t_start=T - 20, t_stop=T - 40, **skw)
torch.cuda.synchronize()
