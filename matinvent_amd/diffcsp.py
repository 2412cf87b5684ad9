"""DiffCSP diffusion module on the HIP path.

Mirror of models/diffcsp/diffusion.py:69-399 (DiffCSPModule): same constructor hparams,
same buffers (`beta_scheduler.*`, `sigma_scheduler.*`), same methods
`sample / add_noise / calc_sample_loss / calc_kl_reg`, same `decoder.*` parameter names in
`state_dict()`.  The arithmetic runs in libmatinvent_hip.so; there is no PyTorch fallback.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .cspnet import CSPNet, CrystalBatch, MAX_ATOMIC_NUM, _ptr, _stream
from .schedules import BetaScheduler, SigmaScheduler, sampler_coefficients, time_embedding_freqs


def _cfg(d, drop=("_target_",)):
    return {k: v for k, v in dict(d).items() if k not in drop}


class SinusoidalTimeEmbeddings(nn.Module):
    """diffusion.py:53-66, evaluated on the device from the host-built frequency table."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.register_buffer("freqs", time_embedding_freqs(dim), persistent=False)

    def forward(self, time):
        lib = _lib.load()
        times = time.to(self.freqs.device, torch.int32).contiguous()
        out = torch.empty(times.shape[0], self.dim, device=self.freqs.device)
        _lib.check(lib.mi_time_embedding(_ptr(times), _ptr(self.freqs), times.shape[0], self.dim, _ptr(out), _stream()))
        return out


class DiffCSPModule(nn.Module):
    def __init__(self, decoder, beta_scheduler, sigma_scheduler, latent_dim=0, time_dim=256, cost_lattice=1.0,
                 cost_coord=1.0, cost_type=20.0, device=None, **kwargs):
        super().__init__()
        self.hparams = dict(decoder=dict(decoder), beta_scheduler=dict(beta_scheduler), sigma_scheduler=dict(sigma_scheduler),
                            latent_dim=latent_dim, time_dim=time_dim, cost_lattice=cost_lattice, cost_coord=cost_coord,
                            cost_type=cost_type, **kwargs)
        dev = torch.device(device if device is not None else "cuda")
        # diffusion.py:73 -- latent_dim + time_dim, pred_type=True, smooth=True
        self.decoder = CSPNet(**_cfg(decoder), latent_dim=latent_dim + time_dim, pred_type=True, smooth=True, device=dev)
        self.beta_scheduler = BetaScheduler(**_cfg(beta_scheduler))
        self.sigma_scheduler = SigmaScheduler(**_cfg(sigma_scheduler))
        self.time_dim = time_dim
        self.time_embedding = SinusoidalTimeEmbeddings(time_dim)
        self.cost_lattice, self.cost_coord, self.cost_type = cost_lattice, cost_coord, cost_type
        self.keep_lattice = cost_lattice < 1e-5
        self.keep_coords = cost_coord < 1e-5
        if self.keep_lattice or self.keep_coords:
            raise NotImplementedError("keep_lattice / keep_coords (CSP mode) are not on the HIP path")
        self.to(dev)

    @property
    def device(self):
        return self.decoder.theta.device

    def _coefficients(self, step_lr):
        """Per-step scalar table, cached until a scheduler buffer changes (checkpoint load)."""
        key = (float(step_lr),) + tuple(b._version for b in (self.beta_scheduler.alphas, self.beta_scheduler.alphas_cumprod,
                                                              self.beta_scheduler.sigmas, self.sigma_scheduler.sigmas,
                                                              self.sigma_scheduler.sigmas_norm))
        cache = self.__dict__.setdefault("_coef_cache", {})
        if key not in cache:
            cache.clear()
            cache[key] = sampler_coefficients(self.beta_scheduler, self.sigma_scheduler, step_lr).contiguous()
        return cache[key]

    def crystal_batch(self, batch, node_offset=0, graph_offset=0) -> CrystalBatch:
        """Index tables for `batch` (anything with .num_atoms); cached on the object."""
        cb = getattr(batch, "_mi_batch", None)
        if cb is None or cb.num_atoms_list != [int(x) for x in batch.num_atoms.tolist()]:
            cb = self.decoder.make_batch(batch.num_atoms, node_offset, graph_offset)
            try:
                batch._mi_batch = cb
            except AttributeError:
                pass
        return cb

    @torch.no_grad()
    def sample(self, batch, diff_ratio=1.0, step_lr=1e-5, seed=0, noise=None, init=None, record=False, t_start=None,
               t_stop=0, node_offset=0, graph_offset=0):
        """DiffCSPModule.sample (diffusion.py:273-399).

        Returns (traj[t_stop], traj) like the reference.  `traj` holds every step only when
        record=True (the reference keeps all T+1 states alive on the device); otherwise it
        holds the final state.  Noise comes from the library's counter-based Philox stream
        keyed by `seed` unless `noise` supplies per-step arrays (dict with corr_x/pred_l/
        pred_t/pred_x: [T+1, ...] tensors) and `init` the initial state (x_T, l_T, t_T).
        """
        lib = _lib.load()
        dev = self.device
        T = self.beta_scheduler.timesteps
        t_start = T if t_start is None else t_start
        cb = batch if isinstance(batch, CrystalBatch) else self.crystal_batch(batch, node_offset, graph_offset)
        B, N = cb.num_graphs, cb.num_nodes
        self.decoder.sync()
        if init is None:
            x = torch.empty(N, 3, device=dev)
            l = torch.empty(B, 3, 3, device=dev)
            a = torch.empty(N, MAX_ATOMIC_NUM, device=dev)
            _lib.check(lib.mi_sampler_init_state(cb._h, seed, T, _ptr(a), _ptr(x), _ptr(l), _stream()), "mi_sampler_init_state")
        else:
            x, l, a = (v.to(dev, torch.float32).contiguous().clone() for v in init)
        x = x % 1.0  # traj[T]['frac_coords'] = x_T % 1 (diffusion.py:289)
        coef = self._coefficients(step_lr)
        nz = None
        if noise is not None:
            keep = {k: noise[k].to(dev, torch.float32).contiguous() for k in ("corr_x", "pred_l", "pred_t", "pred_x")}
            nz = _lib.SamplerNoise(*(keep[k].data_ptr() for k in ("corr_x", "pred_l", "pred_t", "pred_x")))
        rec, rec_t = None, None
        if record:
            z = lambda *s: torch.zeros(*s, device=dev)
            rec_t = dict(atom_types=z(T + 1, N, MAX_ATOMIC_NUM), frac_coords=z(T + 1, N, 3), lattices=z(T + 1, B, 3, 3),
                         frac_coords_mid=z(T + 1, N, 3), log_prob_l=z(T + 1, B), log_prob_t=z(T + 1, B), log_prob_x=z(T + 1, B))
            rec = _lib.SamplerRecord(*(rec_t[k].data_ptr() for k in ("atom_types", "frac_coords", "lattices", "frac_coords_mid",
                                                                     "log_prob_l", "log_prob_t", "log_prob_x")))
        _lib.check(lib.mi_sampler_run(self.decoder._h, cb._h, coef.numpy().ctypes.data_as(C.POINTER(C.c_float)), T, t_start, t_stop,
                                      _ptr(self.time_embedding.freqs), seed, C.byref(nz) if nz is not None else None,
                                      C.byref(rec) if rec is not None else None, _ptr(a), _ptr(x), _ptr(l), _stream()),
                   "mi_sampler_run")
        final = dict(atom_types=a, frac_coords=x, lattices=l, num_atoms=cb.num_atoms, batch_idx=cb.batch)
        traj = {t_stop: final}
        if record:
            for t in range(t_start, t_stop - 1, -1):
                d = dict(atom_types=rec_t["atom_types"][t], frac_coords=rec_t["frac_coords"][t], lattices=rec_t["lattices"][t],
                         num_atoms=cb.num_atoms, batch_idx=cb.batch)
                if t > max(t_stop, 1):
                    d.update(log_prob_l=rec_t["log_prob_l"][t], log_prob_t=rec_t["log_prob_t"][t],
                             log_prob_x=rec_t["log_prob_x"][t], frac_coords_mid=rec_t["frac_coords_mid"][t])
                traj[t] = d
        return final, traj
