#!/usr/bin/env python
"""Generate golden vectors by running the reference's own vendored DiffCSP code.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  The reference modules are imported through tests/oracle_shims (pure-torch
stand-ins for torch_scatter / PyG / lightning / hydra); noise is injected by
monkey-patching the torch RNG entry points the reference calls, so the very same noise
arrays can later be fed to the oracle and to the HIP path.

Outputs (committed): tests/golden/*.npz  -- data only (inputs + expected outputs).
Usage: python tests/golden/make_golden.py
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests", "oracle_shims"))
sys.path.insert(0, "/root/reference")
# `models/diffcsp/sample.py` imports pipeline.filters (-> mattergen); give it an empty module.
_pf = types.ModuleType("pipeline.filters")
_pf.invalid_filter = None
_pk = types.ModuleType("pipeline")
_pk.filters = _pf
_pk.__path__ = []
sys.modules["pipeline"] = _pk
sys.modules["pipeline.filters"] = _pf

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)

from models.diffcsp import cspnet as ref_cspnet  # noqa: E402
from models.diffcsp import diffusion as ref_diffusion  # noqa: E402
from models.diffcsp import sample as ref_sample  # noqa: E402
from models.diffcsp import scheduler as ref_sched  # noqa: E402
from models.diffcsp import utils as ref_utils  # noqa: E402
from torch_geometric.data import Data  # noqa: E402  (shim)


def time_freqs(dim):
    """diffusion.py:61-63 evaluated here (the module recomputes it on every forward)."""
    import math
    half = dim // 2
    return torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays)")


class NoiseTape:
    """Serves pre-generated arrays to torch.rand / randn / randn_like in call order."""

    def __init__(self, items):
        self.items = list(items)
        self.i = 0

    def _next(self, shape):
        a = self.items[self.i]
        self.i += 1
        assert tuple(a.shape) == tuple(shape), (self.i, a.shape, shape)
        return a.clone()

    def rand(self, shape, *a, **k):
        return self._next(shape)

    def randn(self, shape, *a, **k):
        return self._next(shape)

    def randn_like(self, x, *a, **k):
        return self._next(x.shape)

    def __enter__(self):
        self._saved = (torch.rand, torch.randn, torch.randn_like)
        torch.rand, torch.randn, torch.randn_like = self.rand, self.randn, self.randn_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn, torch.randn_like = self._saved
        if exc[0] is None:
            assert self.i == len(self.items), (self.i, len(self.items))


def make_module(H, L, F, T, seed, sigmas_norm_seed=1234, ln=True, edge_style="fc", head_scale=1.0, cost_lattice=1.0, cost_coord=1.0):
    hp = dict(
        decoder=dict(_target_="models.diffcsp.cspnet.CSPNet", hidden_dim=H, num_layers=L, max_atoms=100,
                     act_fn="silu", dis_emb="sin", num_freqs=F, edge_style=edge_style, ln=ln, ip=True,
                     cutoff=7.0, max_neighbors=20),
        beta_scheduler=dict(_target_="models.diffcsp.scheduler.BetaScheduler", timesteps=T, scheduler_mode="cosine"),
        sigma_scheduler=dict(_target_="models.diffcsp.scheduler.SigmaScheduler", timesteps=T, sigma_begin=0.005,
                             sigma_end=0.5),
        latent_dim=0, time_dim=256, cost_lattice=cost_lattice, cost_coord=cost_coord, cost_type=20.0)
    # reference construction order: decoder (weights) first, then the schedulers
    # (SigmaScheduler draws 10000*T randn for sigmas_norm, scheduler.py:46-51,109)
    torch.manual_seed(seed)
    dec_only = ref_cspnet.CSPNet(hidden_dim=H, num_layers=L, max_atoms=100, num_freqs=F, edge_style=edge_style,
                                 ln=ln, ip=True, cutoff=7.0, max_neighbors=20, latent_dim=256, pred_type=True,
                                 smooth=True)
    torch.manual_seed(sigmas_norm_seed)
    m = ref_diffusion.DiffCSPModule(**hp)
    m.decoder.load_state_dict(dec_only.state_dict())
    if head_scale != 1.0:
        with torch.no_grad():
            m.decoder.coord_out.weight *= head_scale
            m.decoder.lattice_out.weight *= head_scale
            m.decoder.type_out.weight *= head_scale
            m.decoder.type_out.bias *= head_scale
    m.eval()
    return m


def batch_of(num_atoms):
    na = torch.as_tensor(num_atoms, dtype=torch.long)
    B, N = len(na), int(na.sum())
    return Data(num_atoms=na, num_graphs=B, num_nodes=N, batch=torch.repeat_interleave(torch.arange(B), na))


def ft_batch(num_atoms, gen):
    b = batch_of(num_atoms)
    B, N = b.num_graphs, b.num_nodes
    b.lengths = 4.0 + 6.0 * torch.rand(B, 3, generator=gen)
    b.angles = 70.0 + 40.0 * torch.rand(B, 3, generator=gen)
    b.frac_coords = torch.rand(N, 3, generator=gen)
    b.atom_types = torch.randint(1, 95, (N,), generator=gen)
    b.reward = torch.rand(B, generator=gen)
    return b


# -------------------------------------------------------------------------------------------
def g1_repeat_blocks():
    """(1) the seven docstring KATs of repeat_blocks (utils.py:208-226), outputs from the code."""
    cases = [
        dict(sizes=[1, 3, 2], repeats=[3, 2, 3], continuous_indexing=False),
        dict(sizes=[1, 3, 2], repeats=[3, 2, 3], continuous_indexing=True),
        dict(sizes=[1, 3, 2], repeats=[3, 2, 3], continuous_indexing=True, repeat_inc=4),
        dict(sizes=[1, 3, 2], repeats=[3, 2, 3], continuous_indexing=True, start_idx=5),
        dict(sizes=[1, 3, 2], repeats=[3, 2, 3], continuous_indexing=True, block_inc=1),
        dict(sizes=[0, 3, 2], repeats=[3, 2, 3], continuous_indexing=True),
        dict(sizes=[2, 3, 2], repeats=[2, 0, 2], continuous_indexing=True),
    ]
    docstring = [
        [0, 0, 0, 0, 1, 2, 0, 1, 2, 0, 1, 0, 1, 0, 1],
        [0, 0, 0, 1, 2, 3, 1, 2, 3, 4, 5, 4, 5, 4, 5],
        [0, 4, 8, 1, 2, 3, 5, 6, 7, 4, 5, 8, 9, 12, 13],
        [5, 5, 5, 6, 7, 8, 6, 7, 8, 9, 10, 9, 10, 9, 10],
        [0, 0, 0, 2, 3, 4, 2, 3, 4, 6, 7, 6, 7, 6, 7],
        [0, 1, 2, 0, 1, 2, 3, 4, 3, 4, 3, 4],
        [0, 1, 0, 1, 5, 6, 5, 6],
    ]
    out = {}
    for i, c in enumerate(cases):
        kw = {k: v for k, v in c.items() if k not in ("sizes", "repeats")}
        r = ref_utils.repeat_blocks(torch.tensor(c["sizes"]), torch.tensor(c["repeats"]), **kw)
        assert r.tolist() == docstring[i], (i, r.tolist(), docstring[i])
        out[f"c{i}_sizes"] = c["sizes"]
        out[f"c{i}_repeats"] = c["repeats"]
        out[f"c{i}_flags"] = [int(c.get("continuous_indexing", True)), c.get("start_idx", 0),
                              c.get("block_inc", 0), c.get("repeat_inc", 0)]
        out[f"c{i}_out"] = r
    # scalar-repeat form used by reorder_symmetric_edges (cspnet.py:211-216)
    r = ref_utils.repeat_blocks(torch.tensor([2, 0, 3, 1]), repeats=2, continuous_indexing=True, repeat_inc=6)
    out["c7_sizes"], out["c7_repeats"], out["c7_flags"], out["c7_out"] = [2, 0, 3, 1], [2], [1, 0, 0, 6], r
    npz("g1_repeat_blocks", **out)


def g2_schedulers():
    out = {}
    for T in (20, 1000):
        b = ref_sched.BetaScheduler(timesteps=T, scheduler_mode="cosine")
        torch.manual_seed(1234)
        s = ref_sched.SigmaScheduler(timesteps=T, sigma_begin=0.005, sigma_end=0.5)
        for k in ("betas", "alphas", "alphas_cumprod", "sigmas"):
            out[f"T{T}_beta_{k}"] = getattr(b, k)
        out[f"T{T}_sigma_sigmas"] = s.sigmas
        out[f"T{T}_sigma_sigmas_norm"] = s.sigmas_norm
    for mode in ("linear", "quadratic", "sigmoid"):
        b = ref_sched.BetaScheduler(timesteps=50, scheduler_mode=mode)
        out[f"T50_{mode}_betas"] = b.betas
        out[f"T50_{mode}_sigmas"] = b.sigmas
    # wrapped-normal helpers on on-distribution points
    g = torch.Generator().manual_seed(5)
    sig = torch.tensor([0.005, 0.02, 0.1, 0.3, 0.5])[None, :].repeat(64, 1)
    x = (sig * torch.randn(sig.shape, generator=g)) % 1.0
    out["wn_x"], out["wn_sigma"] = x, sig
    out["wn_dlogp"] = ref_sched.d_log_p_wrapped_normal(x, sig)
    out["wn_p"] = ref_sched.p_wrapped_normal(x, sig)
    mu = torch.rand(sig.shape, generator=g)
    xx = (mu + sig * torch.randn(sig.shape, generator=g)) % 1.0
    out["lp_x"], out["lp_mu"] = xx, mu
    out["lp_out"] = ref_diffusion.log_prob_wn(xx, mu, sig)
    npz("g2_schedulers", **out)


def g3_lattice():
    g = torch.Generator().manual_seed(3)
    lengths = 3.0 + 9.0 * torch.rand(8, 3, generator=g)
    angles = 60.0 + 60.0 * torch.rand(8, 3, generator=g)
    angles[0] = torch.tensor([90.0, 90.0, 90.0])
    angles[1] = torch.tensor([90.0, 90.0, 120.0])
    mat = ref_utils.lattice_params_to_matrix_torch(lengths, angles)
    l2, a2 = ref_sample.lattices_to_params_shape(mat)
    rnd = torch.randn(8, 3, 3, generator=g)
    l3, a3 = ref_sample.lattices_to_params_shape(rnd)
    npz("g3_lattice", lengths=lengths, angles=angles, matrix=mat, back_lengths=l2, back_angles=a2,
        rnd=rnd, rnd_lengths=l3, rnd_angles=a3)


def g4_embeddings():
    g = torch.Generator().manual_seed(4)
    x = torch.rand(10, 3, generator=g)
    out = {"x": x}
    for F in (8, 128):
        out[f"sin_F{F}"] = ref_cspnet.SinusoidsEmbedding(n_frequencies=F)(x)
    t = torch.tensor([1, 2, 17, 500, 999, 1000])
    out["t"] = t
    out["time_256"] = ref_diffusion.SinusoidalTimeEmbeddings(256)(t)
    # the frequency table as evaluated on the generating machine (torch.exp is libm-dependent:
    # another host's CPU differs by 1 ulp in a few entries, which t <= 1000 amplifies to 6e-5)
    out["time_freqs_256"] = time_freqs(256)
    npz("g4_embeddings", **out)


def _net_inputs(m, num_atoms, gen, t=7, lat_scale=1.0):
    b = batch_of(num_atoms)
    B, N = b.num_graphs, b.num_nodes
    t_emb = m.time_embedding(torch.full((B,), t))
    at = torch.randn(N, 100, generator=gen)
    fr = torch.rand(N, 3, generator=gen)
    lat = lat_scale * torch.randn(B, 3, 3, generator=gen)
    return b, t_emb, at, fr, lat


def g5_cspnet():
    # (5a) tiny net, all weights stored, per-layer taps
    m = make_module(H=64, L=2, F=8, T=20, seed=0)
    g = torch.Generator().manual_seed(50)
    b, t_emb, at, fr, lat = _net_inputs(m, [3, 5, 1, 7], g)
    taps = {}
    hooks = []
    for l in range(2):
        layer = m.decoder._modules[f"csp_layer_{l}"]
        hooks.append(layer.register_forward_hook(lambda mod, i, o, l=l: taps.__setitem__(f"h_{l}", o.detach())))
    with torch.no_grad():
        pl, px, pt = m.decoder(t_emb, at, fr, lat, b.num_atoms, b.batch)
        edges, fd = m.decoder.gen_edges(b.num_atoms, fr, lat, b.batch)
    for h in hooks:
        h.remove()
    sd = {"P__" + k: v for k, v in m.state_dict().items() if k.startswith("decoder.")}
    npz("g5a_cspnet_tiny", num_atoms=b.num_atoms, t_emb=t_emb, atom_types=at, frac=fr, lattices=lat,
        pred_l=pl, pred_x=px, pred_t=pt, edges=edges, frac_diff=fd, **taps, **sd)

    # (5b) north-star hparams; weights reproducible from manual_seed(0) construction (checksums stored)
    m = make_module(H=512, L=6, F=128, T=20, seed=0)
    g = torch.Generator().manual_seed(51)
    b, t_emb, at, fr, lat = _net_inputs(m, [3, 20], g)
    with torch.no_grad():
        pl, px, pt = m.decoder(t_emb, at, fr, lat, b.num_atoms, b.batch)
    sd = m.state_dict()
    names = [k for k in sd if k.startswith("decoder.")]
    sums = np.array([float(sd[k].double().sum()) for k in names])
    abss = np.array([float(sd[k].double().abs().sum()) for k in names])
    npz("g5b_cspnet_ns", num_atoms=b.num_atoms, t_emb=t_emb, atom_types=at, frac=fr, lattices=lat,
        pred_l=pl, pred_x=px, pred_t=pt, param_names=np.array(names), param_sum=sums, param_abs_sum=abss,
        n_params=np.array(sum(p.numel() for p in m.parameters())),
        w_sample=sd["decoder.csp_layer_3.edge_mlp.0.weight"][:4, :16])


def g5c_knn():
    """knn branch of gen_edges on sane cells (cspnet.py:243-257; utils.py:335-601)."""
    m = make_module(H=64, L=2, F=8, T=20, seed=0, edge_style="knn")
    g = torch.Generator().manual_seed(52)
    num_atoms = [4, 9, 20, 2]
    b = batch_of(num_atoms)
    B, N = b.num_graphs, b.num_nodes
    lengths = 4.0 + 4.0 * torch.rand(B, 3, generator=g)
    angles = 75.0 + 30.0 * torch.rand(B, 3, generator=g)
    lat = ref_utils.lattice_params_to_matrix_torch(lengths, angles)
    fr = torch.rand(N, 3, generator=g)
    at = torch.randn(N, 100, generator=g)
    t_emb = m.time_embedding(torch.full((B,), 5))
    with torch.no_grad():
        cart = torch.einsum("bi,bij->bj", fr, lat[b.batch])
        ei, jim, nb = ref_utils.radius_graph_pbc(cart, None, None, b.num_atoms, 7.0, 20, device="cpu", lattices=lat)
        edges, vec = m.decoder.gen_edges(b.num_atoms, fr, lat, b.batch)
        pl, px, pt = m.decoder(t_emb, at, fr, lat, b.num_atoms, b.batch)
    sd = {"P__" + k: v for k, v in m.state_dict().items() if k.startswith("decoder.")}
    npz("g5c_knn", num_atoms=b.num_atoms, lattices=lat, frac=fr, atom_types=at, t_emb=t_emb,
        rg_edge_index=ei, rg_to_jimages=jim, rg_num_bonds=nb, edges=edges, edge_vec=vec,
        pred_l=pl, pred_x=px, pred_t=pt, **sd)


def sampler_tape(noise, T, N, B):
    items = [noise["x_T"], noise["l_T"], noise["t_T"]]
    for t in range(T, 1, -1):
        items += [torch.zeros(B, 3, 3), torch.zeros(N, 100), noise["corr_x"][t],
                  noise["pred_l"][t], noise["pred_t"][t], noise["pred_x"][t]]
    return NoiseTape(items)


def g6_sample():
    T = 20
    m = make_module(H=64, L=2, F=8, T=T, seed=0)
    num_atoms = [3, 5]
    b = batch_of(num_atoms)
    B, N = b.num_graphs, b.num_nodes
    g = torch.Generator().manual_seed(60)
    noise = dict(x_T=torch.rand(N, 3, generator=g), l_T=torch.randn(B, 3, 3, generator=g),
                 t_T=torch.randn(N, 100, generator=g), corr_x={}, pred_l={}, pred_t={}, pred_x={})
    for t in range(T, 1, -1):
        noise["corr_x"][t] = torch.randn(N, 3, generator=g)
        noise["pred_l"][t] = torch.randn(B, 3, 3, generator=g)
        noise["pred_t"][t] = torch.randn(N, 100, generator=g)
        noise["pred_x"][t] = torch.randn(N, 3, generator=g)
    with sampler_tape(noise, T, N, B):
        final, traj = m.sample(b, step_lr=5e-6)
    out = dict(num_atoms=b.num_atoms, T=np.array(T), step_lr=np.array(5e-6), time_freqs=time_freqs(256),
               x_T=noise["x_T"], l_T=noise["l_T"], t_T=noise["t_T"])
    for t in range(T, 1, -1):
        for k in ("corr_x", "pred_l", "pred_t", "pred_x"):
            out[f"n_{k}_{t}"] = noise[k][t]
    for t in range(T, -1, -1):
        for k in ("atom_types", "frac_coords", "lattices"):
            out[f"traj_{t}_{k}"] = traj[t][k]
        if t > 1:
            for k in ("log_prob_l", "log_prob_t", "log_prob_x", "frac_coords_mid"):
                out[f"traj_{t}_{k}"] = traj[t][k]
    sd = {"P__" + k: v for k, v in m.state_dict().items()}
    npz("g6_sample", **out, **sd)


def g7_noise_loss():
    T = 1000
    m = make_module(H=64, L=2, F=8, T=T, seed=0)
    g = torch.Generator().manual_seed(70)
    b = ft_batch([4, 2, 6], g)
    B, N = b.num_graphs, b.num_nodes
    prior = make_module(H=64, L=2, F=8, T=T, seed=1)
    out = dict(num_atoms=b.num_atoms, lengths=b.lengths, angles=b.angles, frac_coords=b.frac_coords,
               atom_types=b.atom_types, sigmas_norm=m.sigma_scheduler.sigmas_norm)
    for ti in (0, 500, 999):
        rl, rx, rt = torch.randn(B, 3, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, 100, generator=g)
        with NoiseTape([rl, rx, rt]):
            noised = m.add_noise(b, ti)
        (t_emb, atp, ifr, ilat, na, bb), (rand_l, tar_x, rand_t), _ = noised
        with torch.no_grad():
            loss, pred = m.calc_sample_loss(noised)
            _, ppred = prior.calc_sample_loss(noised)
            kl = m.calc_kl_reg(pred, ppred, b)
        out.update({f"t{ti}_rand_l": rl, f"t{ti}_rand_x": rx, f"t{ti}_rand_t": rt, f"t{ti}_t_emb": t_emb,
                    f"t{ti}_atom_type_probs": atp, f"t{ti}_input_frac": ifr, f"t{ti}_input_lattice": ilat,
                    f"t{ti}_tar_x": tar_x, f"t{ti}_loss": loss, f"t{ti}_kl": kl,
                    f"t{ti}_pred_l": pred[0], f"t{ti}_pred_x": pred[1], f"t{ti}_pred_t": pred[2]})
    sd = {"P__" + k: v for k, v in m.state_dict().items() if k.startswith("decoder.")}
    sdp = {"Q__" + k: v for k, v in prior.state_dict().items() if k.startswith("decoder.")}
    npz("g7_noise_loss", **out, **sd, **sdp)


def g8_ft_step():
    """3 fine-tune micro-steps + 1 Adam step, accum=3, following pipeline/mat_invent.py:136-177
    literally over the imported model methods (the pipeline module itself is not importable)."""
    T = 1000
    agent = make_module(H=64, L=2, F=8, T=T, seed=0)
    prior = make_module(H=64, L=2, F=8, T=T, seed=0)
    g = torch.Generator().manual_seed(80)
    b = ft_batch([4, 2, 6, 3], g)
    B, N = b.num_graphs, b.num_nodes
    lr, accum, sigma, timesteps = 1e-4, 3, 0.025, 6
    for p in agent.parameters():
        p.requires_grad = True
    for p in prior.parameters():
        p.requires_grad = False
    # perturb the agent a little so the anchor penalty is non-zero
    with torch.no_grad():
        for p in agent.parameters():
            p.add_(0.01 * torch.randn(p.shape, generator=g))
    out = dict(num_atoms=b.num_atoms, lengths=b.lengths, angles=b.angles, frac_coords=b.frac_coords,
               atom_types=b.atom_types, reward=b.reward, lr=np.array(lr), accum=np.array(accum),
               sigma=np.array(sigma), timesteps=np.array(timesteps),
               sigmas_norm=agent.sigma_scheduler.sigmas_norm)
    out.update({"P__" + k: v.detach().clone() for k, v in agent.state_dict().items() if k.startswith("decoder.")})
    out.update({"Q__" + k: v.detach().clone() for k, v in prior.state_dict().items() if k.startswith("decoder.")})
    optimizer = torch.optim.Adam(agent.parameters(), lr=lr)
    agent.train()
    optimizer.zero_grad()
    nstep = 0
    for t in range(timesteps):
        rl, rx, rt = torch.randn(B, 3, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, 100, generator=g)
        out.update({f"s{t}_rand_l": rl, f"s{t}_rand_x": rx, f"s{t}_rand_t": rt})
        with NoiseTape([rl, rx, rt]):
            noised = agent.add_noise(b, t)
        sample_loss, agent_pred = agent.calc_sample_loss(noised)
        _, prior_pred = prior.calc_sample_loss(noised)
        adv = b.reward
        _loss_diff = adv * sample_loss
        kl_term = agent.calc_kl_reg(agent_pred, prior_pred, b)
        _loss_kl = kl_term * (1.1 - b.reward)
        _loss = (_loss_diff + _loss_kl * sigma).mean() / accum
        _loss.backward()
        out.update({f"s{t}_sample_loss": sample_loss.detach(), f"s{t}_kl": kl_term.detach(), f"s{t}_loss": _loss.detach()})
        if (t + 1) % accum == 0:
            for k, p in agent.named_parameters():
                if k.startswith("decoder."):
                    out[f"G{nstep}__" + k] = p.grad.detach().clone()
            optimizer.step()
            optimizer.zero_grad()
            for k, p in agent.named_parameters():
                if k.startswith("decoder."):
                    out[f"A{nstep}__" + k] = p.detach().clone()
            nstep += 1
    npz("g8_ft_step", **out)


def g9_host_glue():
    np.random.seed(0)
    ds = ref_sample.SampleDataset(total_num=8)
    np.random.seed(0)
    ds64 = ref_sample.SampleDataset(total_num=64)
    npz("g9_host_glue", atom_dist_mp20=np.array(ref_sample.ATOM_DIST["mp_20"]),
        step_lr_gen_mp20=np.array(ref_sample.DEFAULT_STEP_LR["gen"]["mp_20"]),
        num_atoms_seed0_8=ds.num_atoms, num_atoms_seed0_64=ds64.num_atoms)


def g10_csp_mode():
    """DiffCSPModule.sample in CSP mode (diffusion.py:78-79, 283-287, 308-312, 330, 348-349): keep_coords (cost_coord = 0) and
    keep_lattice (cost_lattice = 0), T = 10, injected noise."""
    T = 10
    out = {}
    for tag, kw in (("kc", dict(cost_coord=0.0)), ("kl", dict(cost_lattice=0.0))):
        m = make_module(H=64, L=2, F=8, T=T, seed=0, head_scale=0.5, **kw)
        assert m.keep_coords == (tag == "kc") and m.keep_lattice == (tag == "kl")
        g = torch.Generator().manual_seed(100)
        b = ft_batch([4, 6], g)
        B, N = b.num_graphs, b.num_nodes
        noise = dict(x_T=torch.rand(N, 3, generator=g), l_T=torch.randn(B, 3, 3, generator=g),
                     t_T=torch.randn(N, 100, generator=g), corr_x={}, pred_l={}, pred_t={}, pred_x={})
        for t in range(T, 1, -1):
            noise["corr_x"][t] = torch.randn(N, 3, generator=g)
            noise["pred_l"][t] = torch.randn(B, 3, 3, generator=g)
            noise["pred_t"][t] = torch.randn(N, 100, generator=g)
            noise["pred_x"][t] = torch.randn(N, 3, generator=g)
        with sampler_tape(noise, T, N, B):
            final, traj = m.sample(b, step_lr=5e-6)
        out.update({f"{tag}_num_atoms": b.num_atoms, f"{tag}_lengths": b.lengths, f"{tag}_angles": b.angles,
                    f"{tag}_frac_coords": b.frac_coords, f"{tag}_x_T": noise["x_T"], f"{tag}_l_T": noise["l_T"], f"{tag}_t_T": noise["t_T"]})
        for t in range(T, 1, -1):
            for k in ("corr_x", "pred_l", "pred_t", "pred_x"):
                out[f"{tag}_n_{k}_{t}"] = noise[k][t]
        for t in range(T, -1, -1):
            for k in ("atom_types", "frac_coords", "lattices"):
                out[f"{tag}_traj_{t}_{k}"] = traj[t][k]
            if t > 1:
                for k in ("log_prob_l", "log_prob_t", "log_prob_x", "frac_coords_mid"):
                    out[f"{tag}_traj_{t}_{k}"] = traj[t][k]
        if tag == "kc":
            out.update({"P__" + k: v for k, v in m.state_dict().items()})
            out["T"] = np.array(T)
            out["time_freqs"] = time_freqs(256)
    npz("g10_csp_mode", **out)


def g11_noise_sampled_times():
    """DiffCSPModule.add_noise(batch) WITHOUT a time index (diffusion.py:83-84): one uniformly drawn time per crystal from numpy's
    global generator; then calc_sample_loss."""
    T = 1000
    m = make_module(H=64, L=2, F=8, T=T, seed=0)
    g = torch.Generator().manual_seed(110)
    b = ft_batch([4, 2, 6, 5], g)
    B, N = b.num_graphs, b.num_nodes
    rl, rx, rt = torch.randn(B, 3, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, 100, generator=g)
    np.random.seed(11)
    times = np.random.choice(np.arange(1, T + 1), B)
    np.random.seed(11)
    with NoiseTape([rl, rx, rt]):
        noised = m.add_noise(b)
    (t_emb, atp, ifr, ilat, na, bb), (rand_l, tar_x, rand_t), _ = noised
    with torch.no_grad():
        loss, pred = m.calc_sample_loss(noised)
    sd = {"P__" + k: v for k, v in m.state_dict().items() if k.startswith("decoder.")}
    npz("g11_noise_sampled_times", num_atoms=b.num_atoms, lengths=b.lengths, angles=b.angles, frac_coords=b.frac_coords,
        atom_types=b.atom_types, sigmas_norm=m.sigma_scheduler.sigmas_norm, np_seed=np.array(11), times=times,
        rand_l=rl, rand_x=rx, rand_t=rt, t_emb=t_emb, atom_type_probs=atp, input_frac=ifr, input_lattice=ilat, tar_x=tar_x,
        loss=loss, pred_l=pred[0], pred_x=pred[1], pred_t=pred[2], time_freqs=time_freqs(256), **sd)


def g12_mattergen_adapter():
    """The IN-TREE arithmetic of the MatterGen adapter, run from the reference's own files (models/mattergen/pl_module.py, loss.py).  The network and
    the corruptions live in the un-vendored package `mattergen @ 5bb2b397` (env.yml:31) and stay parity-unpinned; what the reference itself computes
    around them does not need that package's arithmetic -- only names to import.  The names are given as EMPTY class stand-ins (no arithmetic of
    ours in them): `DiffusionLightningModule` keeps the diffusion module it is handed, `MaterialsLoss` keeps the weights it is handed and one
    key per included field, `apply` returns the per-field losses the harness planted.  Recorded:
      (a) MatterGenModule.calc_kl_reg (pl_module.py:83-102) on random agent / prior predictions with a ragged batch index;
      (b) MatterGenModule.add_noise's time grid (pl_module.py:55-67): the `t` it hands the corruption, for timestep 0 / 1 / 500 / 998 / 999, T_max 1.0 and 0.8;
      (c) SampleLoss.__call__ (loss.py:36-78): default weights and the weighted stack-sum over planted per-field, per-sample losses."""
    import typing
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        mods[name] = m
        return m

    class BatchedData:   # a type bound only
        pass

    class DiffusionLightningModule(torch.nn.Module):   # upstream: a LightningModule that owns the diffusion module
        def __init__(self, diffusion_module=None, optimizer_partial=None, scheduler_partials=None):
            super().__init__()
            self.diffusion_module = diffusion_module

    class MaterialsLoss:   # upstream builds one loss function per included field and keeps the weights; the functions themselves are upstream arithmetic
        def __init__(self, reduce="sum", d3pm_hybrid_lambda=0.01, include_pos=True, include_cell=True, include_atomic_numbers=True, weights=None):
            self.loss_weights = dict(weights)
            self.d3pm_hybrid_lambda = d3pm_hybrid_lambda
            self.loss_fns = {k: None for k, inc in (("pos", include_pos), ("cell", include_cell), ("atomic_numbers", include_atomic_numbers)) if inc}

    T_ = typing.TypeVar("T_")

    class MultiCorruption(typing.Generic[T_]):
        pass

    planted = {}

    def apply(fns, corruption, x, noisy_x, score_model_output, batch_idx, broadcast, node_is_unmasked):
        assert set(fns) == set(planted) and set(batch_idx) == set(fns)
        return {k: planted[k] for k in fns}   # (dict order = the order of the loss functions, as upstream's apply keeps it)

    mod("omegaconf", DictConfig=dict)
    mod("mattergen")
    mod("mattergen.common")
    mod("mattergen.common.loss", MaterialsLoss=MaterialsLoss)
    mod("mattergen.diffusion")
    mod("mattergen.diffusion.config", Config=object)
    mod("mattergen.diffusion.data")
    mod("mattergen.diffusion.data.batched_data", BatchedData=BatchedData)
    mod("mattergen.diffusion.diffusion_module", DiffusionModule=object)
    mod("mattergen.diffusion.lightning_module", DiffusionLightningModule=DiffusionLightningModule)
    mod("mattergen.diffusion.corruption")
    mod("mattergen.diffusion.corruption.multi_corruption", MultiCorruption=MultiCorruption, apply=apply)
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        import importlib
        pl = importlib.import_module("models.mattergen.pl_module")
        ls = importlib.import_module("models.mattergen.loss")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    gen = torch.Generator().manual_seed(1212)
    out = {}
    # ---- (a) calc_kl_reg ------------------------------------------------------------------------------------------------------------------
    na = torch.tensor([3, 1, 7, 20, 2])
    N, Bn = int(na.sum()), len(na)
    n2g = torch.repeat_interleave(torch.arange(Bn), na)

    class Batch:
        def get_batch_idx(self, key):
            assert key in ("pos", "cell", "atomic_numbers")
            return n2g if key != "cell" else None

        def get_batch_size(self):
            return Bn

    agent = {"pos": torch.randn(N, 3, generator=gen), "cell": torch.randn(Bn, 3, 3, generator=gen), "atomic_numbers": torch.randn(N, 101, generator=gen)}
    prior = {k: v + 0.3 * torch.randn(v.shape, generator=gen) for k, v in agent.items()}

    class Corr:
        T = 1.0

        def sample_marginal(self, batch, t):
            return ("noisy", batch)

    class DM:
        corruption = Corr()

        @staticmethod
        def pre_corruption_fn(b):
            return b

        @staticmethod
        def _get_device(b):
            return torch.device("cpu")
    m = pl.MatterGenModule(diffusion_module=DM())
    kl = m.calc_kl_reg(agent, prior, Batch())
    assert kl.shape == (Bn,)
    out.update(kl_num_atoms=na, kl_agent_pos=agent["pos"], kl_agent_cell=agent["cell"], kl_agent_types=agent["atomic_numbers"],
               kl_prior_pos=prior["pos"], kl_prior_cell=prior["cell"], kl_prior_types=prior["atomic_numbers"], kl_out=kl)
    # ---- (b) the time grid of add_noise -----------------------------------------------------------------------------------------------------
    steps = [0, 1, 500, 998, 999]
    for tmax in (1.0, 0.8):
        DM.corruption.T = tmax
        ts = []
        for k in steps:
            noisy, b2, t = m.add_noise(Batch(), k)
            assert t.shape == (Bn,) and bool((t == t[0]).all()) and noisy == ("noisy", b2)
            ts.append(t[0])
        out["grid_t_Tmax%s" % str(tmax).replace(".", "p")] = torch.stack(ts)
    out["grid_timesteps"] = torch.tensor(steps)
    # ---- (c) SampleLoss ------------------------------------------------------------------------------------------------------------------------
    sl = ls.SampleLoss()
    planted.update({k: torch.rand(Bn, generator=gen) * s for k, s in (("pos", 3.0), ("cell", 0.7), ("atomic_numbers", 5.0))})
    agg, metrics = sl(multi_corruption=types.SimpleNamespace(corruptions=None), batch=Batch(), noisy_batch=None, score_model_output=None, t=torch.zeros(Bn))
    assert agg.shape == (Bn,)
    out.update(loss_fields=np.array(list(sl.loss_fns.keys())), loss_weights=np.array([sl.loss_weights[k] for k in sl.loss_fns]),
               loss_d3pm_hybrid_lambda=np.float64(sl.d3pm_hybrid_lambda), loss_pos=planted["pos"], loss_cell=planted["cell"], loss_types=planted["atomic_numbers"],
               loss_agg=agg, loss_metric_means=torch.stack([metrics[k] for k in sl.loss_fns]))
    # custom weights travel through unchanged
    sl2 = ls.SampleLoss(weights={"atomic_numbers": 2.0, "cell": 0.5, "pos": 0.25})
    agg2, _ = sl2(multi_corruption=types.SimpleNamespace(corruptions=None), batch=Batch(), noisy_batch=None, score_model_output=None, t=torch.zeros(Bn))
    out.update(loss_agg_custom=agg2, loss_weights_custom=np.array([sl2.loss_weights[k] for k in sl2.loss_fns]))
    npz("g12_mattergen_adapter", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:   # only the named fixtures, e.g. `make_golden.py g10_csp_mode g11_noise_sampled_times`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_repeat_blocks()
    g2_schedulers()
    g3_lattice()
    g4_embeddings()
    g5_cspnet()
    g5c_knn()
    g6_sample()
    g7_noise_loss()
    g8_ft_step()
    g9_host_glue()
    g10_csp_mode()
    g11_noise_sampled_times()
    g12_mattergen_adapter()
