"""Benchmark-width forward at 64 crystals: device vs oracle on eight crystals, with unit and with perturbed ScalingFactors, inference and training forwards."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import mattergen_oracle as M
from matinvent_amd.mattergen import MatterGenModule
sys.path.insert(0, "tests")
hp = M.GemNetHParams()
def state(B, seed=41, n=20):
    g = torch.Generator().manual_seed(seed)
    mu = (n / 0.05771451654022283) ** (1 / 3)
    cell = mu * torch.eye(3)[None].repeat(B, 1, 1) + 0.3 * M.symmetric_noise(torch.randn(B, 3, 3, generator=g))
    N = B * n
    return dict(na=torch.full((B,), n, dtype=torch.long), frac=torch.rand(N, 3, generator=g), cell=cell, a=torch.randint(1, 101, (N,), generator=g), t=0.1 + 0.8 * torch.rand(B, generator=g))
s = state(64)
for mode in ("unit", "perturbed-scale-factors-only", "all-perturbed"):
    P = M.init_params(hp, seed=0, head_scale=0.3)
    g = torch.Generator().manual_seed(31)
    for k in P:
        nz = 0.01 * torch.randn(P[k].shape, generator=g)
        if mode == "all-perturbed" or (mode != "unit" and M.is_scale_factor(k)):
            P[k] = P[k] + nz
    m = MatterGenModule(gemnet=dict())
    m.decoder.load_state_dict(P, strict=True)
    gb = m.decoder.make_batch(s["na"])
    with torch.no_grad():
        out = {k: v.clone() for k, v in m.decoder(s["frac"], s["cell"], s["a"], s["t"], gb).items()}
    outT = m.decoder(s["frac"], s["cell"], s["a"], s["t"], gb)   # training forward
    with torch.no_grad():
        ref = M.gemnet_forward(P, hp, s["frac"][:160], s["cell"][:8], s["a"][:160], s["na"][:8], s["t"][:8])
    for k, sl in (("pos", slice(0, 160)), ("cell", slice(0, 8)), ("atomic_numbers", slice(0, 160))):
        r = ref[k].numpy(); sc = np.abs(r).max()
        print(mode, k, "inference err %.2e" % (np.abs(out[k][sl].cpu().numpy() - r).max() / sc), "training-forward err %.2e" % (np.abs(outT[k][sl].detach().cpu().numpy() - r).max() / sc), "max|ref| %.3g" % sc)
    del m, gb
    import gc; gc.collect(); torch.cuda.empty_cache()
