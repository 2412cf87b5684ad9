"""Errors of the LOADED library (MI_LIB_PATH selects it) against the reference-generated fixtures, as one JSON line:
g5a / g5b (score-network outputs), g6 (teacher-forced sampler steps and the free-running 20-step chain, log-probs included), g8 (accumulated
fine-tune gradients and per-sample losses of the first accumulation window).  Every figure is max |error| / max(1, max |reference|) of
its tensor, except where the key says otherwise.  Used by tests/test_gpu_tf32_class.py to state the TF32-class build's tolerance next to
the product library's.  usage (GPU box): [MI_LIB_PATH=...] python -m tests.tools.arith_class_errors"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from matinvent_amd import _lib  # noqa: E402
from tests.gpu_util import Box, load_decoder, params_from_golden, wrap_dist  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
load = lambda name: np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b)
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))


def rel_tensor(a, b):   # relative to the tensor's own largest entry (gradients are small numbers)
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b)
    return float(np.abs(a - b).max()) / max(1e-12, float(np.abs(b).max()))


def forward_errors(out):
    from matinvent_amd.cspnet import CSPNet
    g = load("g5a_cspnet_tiny")
    net = CSPNet(hidden_dim=64, num_layers=2, num_freqs=8, latent_dim=256, ln=True, smooth=True, pred_type=True, device="cuda")
    load_decoder(net, params_from_golden(g))
    T = lambda k: torch.from_numpy(g[k]).cuda()
    with torch.no_grad():
        pl, px, pt = net(T("t_emb"), T("atom_types"), T("frac"), T("lattices"), None, batch=net.make_batch(g["num_atoms"]))
    out["g5a"] = max(rel(pl, g["pred_l"]), rel(px, g["pred_x"]), rel(pt, g["pred_t"]))
    g = load("g5b_cspnet_ns")
    torch.manual_seed(0)
    net = CSPNet(hidden_dim=512, num_layers=6, num_freqs=128, latent_dim=256, ln=True, smooth=True, pred_type=True, device="cuda")
    T = lambda k: torch.from_numpy(g[k]).cuda()
    with torch.no_grad():
        pl, px, pt = net(T("t_emb"), T("atom_types"), T("frac"), T("lattices"), g["num_atoms"])
    out["g5b"] = max(rel(pl, g["pred_l"]), rel(px, g["pred_x"]), rel(pt, g["pred_t"]))


def sampler_errors(out):
    from tests.test_gpu_sampler import _golden_module, _noise
    g = load("g6_sample")
    m, P, T = _golden_module(g)
    na = g["num_atoms"]
    B, N = len(na), int(na.sum())
    noise = _noise(g, T, N, B)
    e_x = e_l = e_t = e_lp = 0.0
    for t in range(T, 0, -1):
        init = tuple(torch.from_numpy(g[f"traj_{t}_{k}"]) for k in ("frac_coords", "lattices", "atom_types"))
        final, traj = m.sample(Box(na), step_lr=float(g["step_lr"]), noise=noise, init=init, record=True, t_start=t, t_stop=t - 1)
        e_x = max(e_x, float(wrap_dist(final["frac_coords"].cpu().numpy(), g[f"traj_{t-1}_frac_coords"]).max()))
        e_l = max(e_l, rel(final["lattices"], g[f"traj_{t-1}_lattices"]))
        e_t = max(e_t, rel(final["atom_types"], g[f"traj_{t-1}_atom_types"]))
        if t > 1:
            for k in ("log_prob_l", "log_prob_t", "log_prob_x"):
                e_lp = max(e_lp, rel(traj[t][k], g[f"traj_{t}_{k}"]))
    out["g6_step"] = dict(frac_wrapped=e_x, lattices=e_l, atom_types=e_t, log_probs=e_lp)
    init = (torch.from_numpy(g["x_T"]), torch.from_numpy(g["l_T"]), torch.from_numpy(g["t_T"]))
    final, traj = m.sample(Box(na), step_lr=float(g["step_lr"]), noise=noise, init=init, record=True)
    e_x = e_l = e_t = e_lp = 0.0
    for t in range(T, -1, -1):
        e_x = max(e_x, float(wrap_dist(traj[t]["frac_coords"].cpu().numpy(), g[f"traj_{t}_frac_coords"]).max()))
        e_l = max(e_l, rel(traj[t]["lattices"], g[f"traj_{t}_lattices"]))
        e_t = max(e_t, rel(traj[t]["atom_types"], g[f"traj_{t}_atom_types"]))
        if t > 1:
            for k in ("log_prob_l", "log_prob_t", "log_prob_x"):
                e_lp = max(e_lp, rel(traj[t][k], g[f"traj_{t}_{k}"]))
    out["g6_chain"] = dict(frac_wrapped=e_x, lattices=e_l, atom_types=e_t, log_probs=e_lp,
                           decoded_types_equal=bool((final["atom_types"].argmax(-1).cpu().numpy() == g["traj_0_atom_types"].argmax(-1)).all()))


def finetune_errors(out):
    from tests.test_gpu_train import T, _ft_setup
    g = load("g8_ft_step")
    agent, prior, batch = _ft_setup(g, 6)
    accum, sigma = int(g["accum"]), float(g["sigma"])
    agent.train()
    agent.decoder.theta.grad = None
    e_loss = 0.0
    for t in range(accum):
        noise = tuple(T(g[f"s{t}_{k}"]) for k in ("rand_l", "rand_x", "rand_t"))
        noised = agent.add_noise(batch, t, noise=noise)
        sample_loss, agent_pred = agent.calc_sample_loss(noised)
        with torch.no_grad():
            _, prior_pred = prior.calc_sample_loss(noised)
        kl = agent.calc_kl_reg(agent_pred, prior_pred, batch)
        loss = (batch.reward * sample_loss + kl * (1.1 - batch.reward) * sigma).mean() / accum
        loss.backward()
        e_loss = max(e_loss, rel_tensor(sample_loss, g[f"s{t}_sample_loss"]), rel_tensor(loss, g[f"s{t}_loss"]))
    worst, name = 0.0, ""
    for k, (o, n, shape) in agent.decoder.layout.items():
        if k == "atom_latent_emb.weight":   # (its time-embedding columns see this host's libm-dependent table: 5e-5 on every path)
            continue
        e = rel_tensor(agent.decoder.theta.grad[o:o + n].view(shape), g["G0__decoder." + k])
        if e > worst:
            worst, name = e, k
    out["g8"] = dict(losses=e_loss, grads_worst_tensor=worst, worst_tensor=name)


if __name__ == "__main__":
    res = {"terms_per_product": int(_lib.load().mi_terms_per_product()), "library": os.path.basename(_lib.LIB_PATH)}
    forward_errors(res)
    sampler_errors(res)
    finetune_errors(res)
    res["saturation_events"] = _lib.saturation_events(reset=True)
    print("ARITH_CLASS_ERRORS " + json.dumps(res))
