"""Per-step latency of a small sampling batch (launch-bound regime): 4 crystals, bench network."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
m = bench.build_module(torch.device("cuda", 0))
na = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "10,7,4,10".split(","))]
class C: num_atoms = torch.tensor(na)
cb = C()
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for streams in (1,):
    m.sample(cb, seed=1, step_lr=5e-6, t_start=1000, t_stop=990, streams=streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.sample(cb, seed=1, step_lr=5e-6, t_start=1000, t_stop=1000 - K, streams=streams)
    torch.cuda.synchronize()
    print(f"atoms {na} streams {streams}: {(time.perf_counter() - t0) * 1e3 / K:.3f} ms per denoising step")
