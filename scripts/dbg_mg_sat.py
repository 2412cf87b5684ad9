import sys, torch
sys.path.insert(0, ".")
from matinvent_amd import _lib
from matinvent_amd.mattergen import MatterGenModule
from oracle import mattergen_oracle as MO
hpd = dict(MO.TINY, emb_atom=128, emb_edge=128, max_neighbors=50, cutoff=7.0)
hp = MO.GemNetHParams(**hpd)
g = torch.Generator().manual_seed(4)
na = torch.tensor([20] * 12); N, B = int(na.sum()), len(na)
frac = torch.rand(N, 3, generator=g)
cell = 6.0 * torch.eye(3)[None].repeat(B, 1, 1) + 0.3 * MO.symmetric_noise(torch.randn(B, 3, 3, generator=g))
a, t = torch.randint(1, 101, (N,), generator=g), 0.1 + 0.8 * torch.rand(B, generator=g)
for name in (None, "int_blocks.0.dense_ca.weight", "edge_emb.weight", "atom_emb.weight", "int_blocks.0.before_skip.0.0.weight", "out_blocks.0.dense_F.weight"):
    P = MO.init_params(hp, seed=2, head_scale=0.5)
    if name: P[name].view(-1)[7] = float("nan")
    m = MatterGenModule(gemnet=hpd); m.decoder.load_state_dict(P, strict=True)
    gb = m.decoder.make_batch(na)
    _lib.saturation_events(reset=True)
    with torch.no_grad():
        out = m.decoder(frac, cell, a, t, gb)
    print(name, "E", gb.graph(frac, cell)["src"].shape[0], "sat", _lib.saturation_events(reset=True), "finite", {k: bool(torch.isfinite(v).all()) for k, v in out.items()})
