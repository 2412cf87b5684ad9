import numpy as np
import torch

from oracle import diffcsp_oracle as O


def params_from_golden(g, prefix="P__"):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith(prefix)}


def load_decoder(net, P):
    """P: dict with reference names ('decoder.x' -> tensor)."""
    sd = {k[len("decoder."):]: v for k, v in P.items() if k.startswith("decoder.")}
    net.load_state_dict(sd, strict=True)
    return net


def make_module(H, L, F, T, P=None, sigmas_norm=None, device="cuda", **kw):
    from matinvent_amd.diffcsp import DiffCSPModule
    if sigmas_norm is None:
        sigmas_norm = torch.ones(T + 1)
    m = DiffCSPModule(decoder=dict(hidden_dim=H, num_layers=L, num_freqs=F, ln=True, edge_style="fc"),
                      beta_scheduler=dict(timesteps=T, scheduler_mode="cosine"),
                      sigma_scheduler=dict(timesteps=T, sigma_begin=0.005, sigma_end=0.5, sigmas_norm=sigmas_norm),
                      device=device, **kw)
    if P is not None:
        load_decoder(m.decoder, P)
    return m


def wrap_dist(a, b):
    d = np.abs(np.asarray(a) - np.asarray(b))
    return np.minimum(d, 1 - d)


class Box:
    def __init__(self, num_atoms):
        self.num_atoms = torch.as_tensor(num_atoms, dtype=torch.long)
