"""Per-crystal sample loss and KL of the fine-tune window test's setup: device vs oracle (which crystals carry the loss difference?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import mattergen_oracle as M
from matinvent_amd.mattergen import MatterGenModule, ChemGraph, ChemGraphBatch, symmetrize_lattice
hp = M.GemNetHParams()
def state(B, seed=41, n=20):
    g = torch.Generator().manual_seed(seed)
    mu = (n / 0.05771451654022283) ** (1 / 3)
    cell = mu * torch.eye(3)[None].repeat(B, 1, 1) + 0.3 * M.symmetric_noise(torch.randn(B, 3, 3, generator=g))
    N = B * n
    return dict(na=torch.full((B,), n, dtype=torch.long), frac=torch.rand(N, 3, generator=g), cell=cell, a=torch.randint(1, 101, (N,), generator=g), t=0.1 + 0.8 * torch.rand(B, generator=g))
B = 64
PERT = os.environ.get("PERT", "all")
P0, Q0 = M.init_params(hp, seed=0, head_scale=0.3), M.init_params(hp, seed=0, head_scale=0.3)
g = torch.Generator().manual_seed(31)
for k in P0:
    nz = 0.01 * torch.randn(P0[k].shape, generator=g)
    if PERT == "all" or (PERT == "noscale" and not M.is_scale_factor(k)):
        P0[k] = P0[k] + nz
agent, prior = MatterGenModule(gemnet=dict()), MatterGenModule(gemnet=dict())
agent.decoder.load_state_dict(P0, strict=True); prior.decoder.load_state_dict(Q0, strict=True)
s = state(B)
na, frac, cell, a = s["na"], s["frac"], s["cell"], s["a"]
N = int(na.sum())
off = [0] + torch.cumsum(na, 0).tolist()
batch = ChemGraphBatch([ChemGraph(frac[off[i]:off[i + 1]], cell[i:i + 1], a[off[i]:off[i + 1]]) for i in range(B)])
rewards = torch.rand(B, generator=g)
corr = M.Corruption()
for ti in range(2):
    nz = (torch.randn(N, 3, generator=g), torch.randn(B, 3, 3, generator=g), torch.rand(N, generator=g))
    with torch.no_grad():
        noised = agent.add_noise(batch, ti, noise=nz)
        loss, pred = agent.calc_sample_loss(noised)
        _, ppred = prior.calc_sample_loss(noised)
        kl = agent.calc_kl_reg(pred, ppred, batch)
    ob = dict(pos=frac, cell=symmetrize_lattice(cell), atomic_numbers=a, num_atoms=na)
    t = torch.full((B,), M.time_grid(corr, ti))
    noisy, aux = M.sample_marginal(corr, ob, t, dict(pos=nz[0], cell=nz[1], types=nz[2]))
    with torch.no_grad():
        pa = M.gemnet_forward(P0, hp, noisy["pos"], noisy["cell"], noisy["atomic_numbers"], na, t)
        pp = M.gemnet_forward(Q0, hp, noisy["pos"], noisy["cell"], noisy["atomic_numbers"], na, t)
    sl, parts = M.sample_loss(corr, ob, aux, pa)
    klo = M.calc_kl_reg(pa, pp, aux["node2graph"], B)
    l, k_ = loss.cpu().numpy(), kl.cpu().numpy()
    dl, dk = np.abs(l - sl.numpy()), np.abs(k_ - klo.numpy())
    print(f"ti {ti}: sample loss sum dev {l.sum():.6g} oracle {float(sl.sum()):.6g} rel {abs(l.sum() - float(sl.sum())) / float(sl.sum()):.2e}; worst crystals {np.argsort(-dl)[:4]} abs diffs {np.sort(dl)[::-1][:4]} of values {sl.numpy()[np.argsort(-dl)[:4]]}")
    print(f"       kl sum dev {k_.sum():.6g} oracle {float(klo.sum()):.6g} rel {abs(k_.sum() - float(klo.sum())) / float(klo.sum()):.2e}; worst {np.argsort(-dk)[:4]} diffs {np.sort(dk)[::-1][:4]} of {klo.numpy()[np.argsort(-dk)[:4]]}")
    for kk in ("pos", "cell", "atomic_numbers"):
        r = pa[kk].numpy(); d = np.abs(pred[kk].cpu().numpy() - r)
        print(f"       pred {kk}: max err {d.max() / np.abs(r).max():.2e} of max|ref| {np.abs(r).max():.3g}")
    gb = agent.decoder.make_batch(na)
    dg = gb.graph(noisy["pos"], noisy["cell"])
    og = M.build_graph(noisy["pos"], noisy["cell"], na, hp)
    ds, dd = dg["src"].cpu().numpy(), dg["dst"].cpu().numpy()
    os_, od = og["src"].numpy(), og["dst"].numpy()
    print("       edges device", len(ds), "oracle", len(os_))
    dev_set = set(zip(ds.tolist(), dd.tolist(), map(tuple, dg["img"].cpu().numpy().tolist()))) if "img" in dg else None
    if dev_set is not None:
        ora_set = set(zip(os_.tolist(), od.tolist(), map(tuple, og["img"].numpy().tolist())))
        diff = dev_set ^ ora_set
        print("       edges in one list only:", len(diff), "crystals:", sorted({d[1] // 20 for d in diff}))
    del gb
