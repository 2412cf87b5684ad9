"""Fine-tune time per timestep on the reference's default-size set (18 crystals, SampleDataset atom counts) as a function of the
number of stacked timesteps (ft_step(stack=...)).  Usage (GPU box): PYTHONPATH=. python scripts/time_stacked_finetune.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from matinvent_amd.data import CrystalData
from matinvent_amd.finetune import ft_step
dev = torch.device('cuda')
agent, prior = bench.build_module(dev), bench.build_module(dev)
prior.requires_grad_(False)
g = torch.Generator().manual_seed(7)
np.random.seed(0)
from matinvent_amd.sampling import SampleDataset
na = SampleDataset(18).num_atoms
data = [CrystalData(torch.rand(int(n), 3, generator=g), torch.randint(1, 95, (int(n),), generator=g), 4 + 6 * torch.rand(1, 3, generator=g), 70 + 40 * torch.rand(1, 3, generator=g)) for n in na]
rewards = torch.rand(18, generator=g).numpy()
cfg = dict(lr=1e-4, accum_steps=50, epochs=1, sigma=0.025)
for stack in (1, 4, 8, 10, 13, 16):
    ft_step(agent, prior, data, rewards, dict(cfg, timesteps=50), log=lambda *_: None, stack=stack); torch.cuda.synchronize()
    t0 = time.perf_counter(); ft_step(agent, prior, data, rewards, dict(cfg, timesteps=200), log=lambda *_: None, stack=stack); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"stack {stack}: {dt*1e3:.2f} ms per timestep")
