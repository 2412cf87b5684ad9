"""MatterGen-shaped model on the HIP path: the surface of models/mattergen/{pl_module,loss,sample,dataset}.py.

PARITY UNPINNED.  The reference adapts the un-vendored package `mattergen @ 5bb2b397` (env.yml:31); its denoiser, corruptions,
loss and sampler cannot be inspected or imported here.  The arithmetic behind this module is the published GemNet-T / MatterGen
description as restated in oracle/mattergen_oracle.py (the GPU tests compare the two); what follows the reference's OWN files:
  * MatterGenModule.add_noise / calc_sample_loss / calc_kl_reg   (pl_module.py:55-102): the time grid
    linspace(T_max, 1/1000, 1000)[timestep], per-sample loss, anchor penalty on pos / cell / atomic_numbers;
  * SampleLoss weights {atomic_numbers 1, cell 1, pos 0.1}       (loss.py:22-26, 71-73);
  * MatterGenSampler.generate returning the `mean` batch          (sample.py:49-50, 270-303);
  * MatterGenDataset.from_samples fields                          (dataset.py:42-65).
There is no CPU fallback: everything numeric runs in libmatinvent_hip.so (csrc/gemnet.hip).
"""
import ctypes as C
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .cspnet import _ptr, _stream
from .data import SimpleStructure, lattices_to_params_shape

NUM_CLASSES, MASK = 101, 101
WEIGHTS = dict(atomic_numbers=1.0, cell=1.0, pos=0.1)  # loss.py:22-26
D3PM_LAMBDA = 0.01                                     # loss.py:15

DEFAULT_GEMNET = dict(emb_atom=512, emb_edge=512, emb_trip=64, emb_rbf=16, emb_cbf=16, emb_bil=64, num_radial=128, num_spherical=7,
                      num_blocks=4, num_before_skip=1, num_after_skip=2, num_concat=1, num_atom=3, max_neighbors=50, max_images=5, cutoff=7.0)
DEFAULT_CORRUPTION = dict(sigma_min=0.005, sigma_max=5.0, beta_min=0.1, beta_max=20.0, limit_density=0.05771451654022283, limit_var_scale=0.25,
                          d3pm_steps=1000)


# ---- records ----------------------------------------------------------------------------------------------------------------------
@dataclass
class ChemGraph:
    """One crystal as the MatterGen side passes it around (mattergen.common.data.chemgraph.ChemGraph): fractional positions,
    3x3 cell, atomic numbers."""
    pos: torch.Tensor              # [n, 3] fractional
    cell: torch.Tensor             # [1, 3, 3]
    atomic_numbers: torch.Tensor   # [n] long
    num_atoms: int = 0
    reward: Optional[torch.Tensor] = None

    def __post_init__(self):
        self.num_atoms = int(self.num_atoms) if self.num_atoms else int(self.pos.shape[0])
        self.cell = self.cell.reshape(1, 3, 3)

    @property
    def atom_types(self):   # the name the DiffCSP-side records use (replay buffer / composition keys)
        return self.atomic_numbers


class ChemGraphBatch:
    """collate() of ChemGraphs: contiguous atoms, `get_batch_idx` = atom -> crystal (the two methods the reference calls,
    pl_module.py:64,94; loss.py:48,58)."""

    def __init__(self, items: List[ChemGraph]):
        self.num_graphs = len(items)
        self.num_atoms = torch.tensor([d.num_atoms for d in items], dtype=torch.long)
        self.batch = torch.repeat_interleave(torch.arange(self.num_graphs), self.num_atoms)
        self.pos = torch.cat([d.pos.float() for d in items]) if items else torch.zeros(0, 3)
        self.cell = torch.cat([d.cell.float().reshape(1, 3, 3) for d in items]) if items else torch.zeros(0, 3, 3)
        self.atomic_numbers = torch.cat([d.atomic_numbers.long() for d in items]) if items else torch.zeros(0, dtype=torch.long)
        if items and all(d.reward is not None for d in items):
            self.reward = torch.cat([torch.as_tensor(d.reward).float().view(1) for d in items])

    def get_batch_idx(self, field=None):
        return self.batch

    def get_batch_size(self):
        return self.num_graphs

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def to_data_list(self):
        off = [0] + torch.cumsum(self.num_atoms, 0).tolist()
        return [ChemGraph(self.pos[off[i]:off[i + 1]].cpu(), self.cell[i:i + 1].cpu(), self.atomic_numbers[off[i]:off[i + 1]].cpu(), int(self.num_atoms[i]))
                for i in range(self.num_graphs)]


def symmetrize_lattice(cell: torch.Tensor) -> torch.Tensor:
    """The per-item transform MatterGenDataset applies before noising (dataset.py:14-16, [UPSTREAM-UNVERIFIED]): replace the cell
    by the symmetric matrix with the same metric, S = sqrtm(L L^T) (same lengths and angles, a rotated frame).  Host glue."""
    L = cell.double().reshape(-1, 3, 3)
    w, U = torch.linalg.eigh(L @ L.transpose(1, 2))
    S = U @ torch.diag_embed(torch.sqrt(w.clamp(min=0))) @ U.transpose(1, 2)
    return S.float().reshape(cell.shape)


class MatterGenDataset:
    """dataset.py:19-65: `from_samples(samples, rewards)` keeps pos / cell / atomic_numbers / num_atoms per crystal, attaches
    `reward`, and applies the transform list (symmetrize_lattice; the chemical-system string is not used on this path)."""

    def __init__(self, items: List[ChemGraph]):
        self.items = items

    @classmethod
    def from_samples(cls, samples, rewards=None, transforms=(symmetrize_lattice,)):
        if isinstance(samples, ChemGraphBatch):
            samples = samples.to_data_list()
        items = []
        for i, s in enumerate(samples):
            cell = s.cell
            for tf in transforms or ():
                cell = tf(cell)
            r = None if rewards is None else torch.tensor([float(np.asarray(rewards)[i])])
            items.append(ChemGraph(s.pos.clone(), cell, s.atomic_numbers.clone(), s.num_atoms, r))
        return cls(items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class ChemGraphLoader:
    def __init__(self, dataset, batch_size, shuffle=True):
        self.dataset, self.batch_size, self.shuffle = dataset, max(1, int(batch_size)), shuffle

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        for s in range(0, n, self.batch_size):
            yield ChemGraphBatch([self.dataset[i] for i in order[s:s + self.batch_size]])


# ---- denoiser ---------------------------------------------------------------------------------------------------------------------
# atoms per chunk of the fine-tune loop (finetune._ft_step_module_surface): the training forward keeps every activation for the
# backward (~33 MB per atom at the benchmark sizes, mirrored by the gradient arena, plus the frozen prior's workspace)
FT_CHUNK_ATOMS = 1280


class GBatch:
    """Graph buffers + activation arena of one batch of crystals (mi_gbatch)."""

    def __init__(self, net: "GemNetTDenoiser", num_atoms, node_offset=0, graph_offset=0):
        lib = _lib.load()
        na = [int(x) for x in (num_atoms.tolist() if torch.is_tensor(num_atoms) else num_atoms)]
        self.num_atoms_list, self.num_graphs, self.num_nodes = na, len(na), sum(na)
        h = C.c_void_p()
        _lib.check(lib.mi_gbatch_create(net._h, (C.c_int * max(len(na), 1))(*na), len(na), node_offset, graph_offset, C.byref(h)), "mi_gbatch_create")
        self._h, self._lib, self._dev, self._net = h, lib, net.theta.device, net
        self.num_atoms = torch.tensor(na, dtype=torch.long, device=self._dev)
        self.batch = torch.repeat_interleave(torch.arange(len(na), device=self._dev), self.num_atoms)

    def graph(self, pos, cell):
        """Build the periodic graph for (pos, cell) and return it: dict(src, dst, img, swap, rowptr, D, V)."""
        f = lambda x: x.detach().to(self._dev, torch.float32).contiguous()
        pos, cell = f(pos), f(cell)
        n = C.c_int64()
        self._net.sync()
        _lib.check(self._lib.mi_gemnet_graph(self._net._h, self._h, _ptr(pos), _ptr(cell), _stream(), C.byref(n)), "mi_gemnet_graph")
        E = int(n.value)
        i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=self._dev)
        out = dict(src=i32(E), dst=i32(E), img=i32(E, 3), swap=i32(E), rowptr=i32(self.num_nodes + 1), D=torch.empty(E, device=self._dev),
                   V=torch.empty(E, 3, device=self._dev))
        _lib.check(self._lib.mi_gemnet_graph_read(self._h, *(_ptr(out[k]) for k in ("src", "dst", "img", "swap", "rowptr", "D", "V")), _stream()),
                   "mi_gemnet_graph_read")
        return {k: (v.long() if v.dtype == torch.int32 else v) for k, v in out.items()}

    def graph_status(self):
        """Per-crystal graph-capacity flags of the forwards since the last chain started (mi_gbatch_graph_status): int32 [B] on the
        CPU, non-zero = the crystal exceeded a capacity of the periodic graph and ran without edges from then on (its sample is
        invalid; the other crystals are unaffected).  Waits for the current stream."""
        out = (C.c_int * max(self.num_graphs, 1))()
        _lib.check(self._lib.mi_gbatch_graph_status(self._h, out, None, _stream()), "mi_gbatch_graph_status")
        return torch.tensor(list(out)[:self.num_graphs], dtype=torch.int32)

    def tap(self, name):
        n = C.c_int64()
        _lib.check(self._lib.mi_gemnet_tap(self._h, name.encode(), None, 0, C.byref(n), _stream()), "mi_gemnet_tap")
        out = torch.empty(int(n.value), device=self._dev)
        _lib.check(self._lib.mi_gemnet_tap(self._h, name.encode(), _ptr(out), out.numel(), None, _stream()), "mi_gemnet_tap")
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and getattr(self, "_lib", None) is not None:
            self._lib.mi_gbatch_destroy(h)


class GemNetFunction(torch.autograd.Function):
    """The denoiser as ONE differentiable op (gradient w.r.t. the flat parameter vector only: the fine-tune step never needs input
    gradients, pipeline/mat_invent.py:164)."""

    @staticmethod
    def forward(ctx, theta, net, gb, pos, cell, types, t):
        lib = _lib.load()
        net.sync()
        dev = theta.device
        B, N = gb.num_graphs, gb.num_nodes
        o_pos, o_cell, o_log = torch.empty(N, 3, device=dev), torch.empty(B, 3, 3, device=dev), torch.empty(N, NUM_CLASSES, device=dev)
        _lib.check(lib.mi_gemnet_forward(net._h, gb._h, _ptr(pos), _ptr(cell), _ptr(types), _ptr(t), _ptr(o_pos), _ptr(o_cell), _ptr(o_log), 1, _stream()),
                   "mi_gemnet_forward")
        ctx.net, ctx.gb, ctx.n = net, gb, theta.numel()
        return o_pos, o_cell, o_log

    @staticmethod
    def backward(ctx, d_pos, d_cell, d_log):
        lib = _lib.load()
        net, gb = ctx.net, ctx.gb
        dev = net.theta.device
        c = lambda g: None if g is None else g.contiguous().float()
        d_pos, d_cell, d_log = c(d_pos), c(d_cell), c(d_log)
        grad = torch.zeros(ctx.n, device=dev)
        _lib.check(lib.mi_gemnet_backward(net._h, gb._h, _ptr(d_pos), _ptr(d_cell), _ptr(d_log), _ptr(grad), _stream()), "mi_gemnet_backward")
        return grad, None, None, None, None, None, None


class GemNetTDenoiser(nn.Module):
    """GemNet-T-shaped denoiser (oracle/mattergen_oracle.py::gemnet_forward): ONE flat fp32 parameter vector in the order of
    mi_gemnet_param_info; `views()` / `state_dict()` expose the named tensors."""

    def __init__(self, device=None, **hp):
        super().__init__()
        cfgd = dict(DEFAULT_GEMNET, **hp)
        self.hp = cfgd
        lib = _lib.load()
        self._lib = lib
        cfg = _lib.GemNetConfig(*[int(cfgd[k]) for k, _ in _lib.GemNetConfig._fields_[:-1]], float(cfgd["cutoff"]))
        h = C.c_void_p()
        _lib.check(lib.mi_gemnet_create(C.byref(cfg), C.byref(h)), "mi_gemnet_create")
        self._h = h
        self.layout = OrderedDict()
        for i in range(lib.mi_gemnet_num_tensors(h)):
            name, off, numel, rows, cols = C.c_char_p(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
            _lib.check(lib.mi_gemnet_param_info(h, i, C.byref(name), C.byref(off), C.byref(numel), C.byref(rows), C.byref(cols)))
            nm = name.value.decode()
            self.layout[nm] = (off.value, numel.value, (cols.value,) if nm.endswith(".bias") else (rows.value, cols.value))
        device = torch.device(device if device is not None else "cuda")
        self.theta = nn.Parameter(torch.zeros(int(lib.mi_gemnet_num_params(h)), dtype=torch.float32, device=device))
        self.theta._mi_owner = self
        self._dirty, self._packed_version = True, -1
        self.reset_parameters()

    def views(self):
        return OrderedDict((k, self.theta.data[o:o + n].view(shape)) for k, (o, n, shape) in self.layout.items())

    @torch.no_grad()
    def reset_parameters(self, head_scale=1.0, cell_head_scale=None):
        """Variance-preserving normal init (std 1/sqrt(fan_in)), zero biases -- random-init runs only (the upstream checkpoints are
        unreachable offline)."""
        for name, w in self.views().items():
            if name.endswith(".bias"):
                w.zero_()
            elif name.endswith(".scale_factor"):   # GemNet's ScalingFactor: a fitted constant upstream (a buffer, never trained); identity here
                w.fill_(1.0)
            elif name == "atom_emb.weight":
                w.copy_(torch.randn(w.shape))
            else:
                w.copy_(torch.randn(w.shape) / math.sqrt(w.shape[1]))
                if ".out_F." in name or ".out_S." in name or ".out_energy." in name or name == "fc_atom.weight":
                    w.mul_(cell_head_scale if (cell_head_scale is not None and ".out_S." in name) else head_scale)
        self._dirty = True

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        for k, w in self.views().items():
            out[prefix + k] = w.detach().clone()
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        with torch.no_grad():
            for k, w in self.views().items():
                if prefix + k in state_dict:
                    w.copy_(state_dict[prefix + k].reshape(w.shape))
                elif strict:
                    missing_keys.append(prefix + k)
        self._dirty = True

    def mark_dirty(self):
        self._dirty = True

    def sync(self):
        if self._dirty or self._packed_version != self.theta._version:
            _lib.check(self._lib.mi_gemnet_set_params(self._h, _ptr(self.theta.data), _stream()), "mi_gemnet_set_params")
            self._dirty, self._packed_version = False, self.theta._version
            self._sync_gen = getattr(self, "_sync_gen", 0) + 1   # (the library rebuilds its weight plane sets lazily after this)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.theta._mi_owner = self
        self._dirty = True
        return r

    def make_batch(self, num_atoms, node_offset=0, graph_offset=0) -> GBatch:
        return GBatch(self, num_atoms, node_offset, graph_offset)

    def forward(self, pos, cell, atomic_numbers, t, batch: GBatch):
        """-> dict(pos [N,3], cell [B,3,3], atomic_numbers [N,101])."""
        dev = self.theta.device
        f = lambda x: x.detach().to(dev, torch.float32).contiguous()
        pos, cell, t = f(pos), f(cell), f(t)
        types = atomic_numbers.detach().to(dev, torch.int32).contiguous()
        B, N = batch.num_graphs, batch.num_nodes
        assert pos.shape == (N, 3) and cell.shape == (B, 3, 3) and t.shape == (B,) and types.shape == (N,)
        if torch.is_grad_enabled() and self.theta.requires_grad:
            o = GemNetFunction.apply(self.theta, self, batch, pos, cell, types, t)
        else:
            self.sync()
            o = (torch.empty(N, 3, device=dev), torch.empty(B, 3, 3, device=dev), torch.empty(N, NUM_CLASSES, device=dev))
            _lib.check(self._lib.mi_gemnet_forward(self._h, batch._h, _ptr(pos), _ptr(cell), _ptr(types), _ptr(t), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), 0,
                                                   _stream()), "mi_gemnet_forward")
        return dict(pos=o[0], cell=o[1], atomic_numbers=o[2])

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and getattr(self, "_lib", None) is not None:
            self._lib.mi_gemnet_destroy(h)


def weighted_field_sum(fields, weights=None):
    """SampleLoss.__call__'s aggregation (loss.py:71-73): stack of w_field * loss_field over the fields in the loss functions' order, summed -> [B]
    (pinned by tests/golden/g12_mattergen_adapter.npz, generated from the reference's own loss.py)."""
    w = WEIGHTS if weights is None else weights
    return torch.stack([w[k] * v for k, v in fields.items()], dim=0).sum(0)


def _scatter_mean(src, index, dim_size):
    out = torch.zeros(dim_size, dtype=src.dtype, device=src.device).index_add(0, index, src)
    cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device).index_add(0, index, torch.ones_like(src))
    return out / cnt.clamp(min=1)


def _d_log_p_wn(x, sigma, N=10):
    num, den = torch.zeros_like(x), torch.zeros_like(x)
    for i in range(-N, N + 1):
        e = torch.exp(-(x + i) ** 2 / 2 / sigma ** 2)
        num = num + (x + i) / sigma ** 2 * e
        den = den + e
    return -num / den


# ---- the module the pipeline talks to ----------------------------------------------------------------------------------------------
class MatterGenModule(nn.Module):
    """Mirror of models/mattergen/pl_module.py::MatterGenModule: add_noise / calc_sample_loss / calc_kl_reg (+ `sample`, which the
    reference reaches through mattergen's PredictorCorrector)."""

    def __init__(self, gemnet=None, corruption=None, device=None, **kwargs):
        super().__init__()
        dev = torch.device(device if device is not None else "cuda")
        self.decoder = GemNetTDenoiser(device=dev, **dict(gemnet or {}))   # (named like the DiffCSP module's network: one ft_step serves both)
        self.corruption = dict(DEFAULT_CORRUPTION, **dict(corruption or {}))
        self.hparams = dict(gemnet=dict(self.decoder.hp), corruption=dict(self.corruption), **kwargs)
        self.T = 1.0
        self.to(dev)

    @property
    def device(self):
        return self.decoder.theta.device

    def _corr(self):
        c = self.corruption
        return _lib.MGCorruption(c["sigma_min"], c["sigma_max"], c["beta_min"], c["beta_max"], c["limit_density"], c["limit_var_scale"], int(c["d3pm_steps"]))

    def collate(self, items, rewards=None):
        """What MatterGenSuite.get_dataloader + MatterGenDataset.from_samples produce for ONE batch holding the whole set
        (pipeline/mat_invent.py:129-133): symmetrised cells, rewards attached."""
        ds = MatterGenDataset.from_samples(items, rewards)
        return ChemGraphBatch([ds[i] for i in range(len(ds))])

    def _chain_batch(self, num_atoms, off, slot):
        """Batch handle of one of several concurrent sampler chains: keyed by (chain slot, atom counts) -- chains of equal shape must not
        share a handle (its arenas are the chain's working set) -- and kept apart from the fine-tune cache's atom budget."""
        key = (int(slot),) + tuple(int(x) for x in num_atoms.tolist())
        cache = self.__dict__.setdefault("_gb_chain_cache", {})
        gb = cache.get(key)
        if gb is None:
            while len(cache) >= 8:
                cache.pop(next(iter(cache)))
            gb = cache[key] = self.decoder.make_batch(list(key[1:]), off[0], off[1])
        _lib.check(gb._lib.mi_gbatch_set_offsets(gb._h, int(off[0]), int(off[1])), "mi_gbatch_set_offsets")
        return gb

    def _batch_for(self, num_atoms):
        """The batch handle (graph buffers + activation arenas) for this atom-count signature, with the current shard / chunk offsets.
        Handles are cached by signature only -- equal-shaped chunks of a large fine-tune set share one handle and its arenas -- and the
        cache is bounded both in entries and in atoms (an arena set is ~100 MB per atom in training)."""
        off = getattr(self, "shard_offsets", (0, 0))
        key = tuple(int(x) for x in num_atoms.tolist())
        cache = self.__dict__.setdefault("_gb_cache", {})
        gb = cache.get(key)
        if gb is None:
            budget = 2 * FT_CHUNK_ATOMS
            while cache and (len(cache) >= 4 or sum(sum(k) for k in cache) + sum(key) > budget):
                cache.pop(next(iter(cache)))
            gb = cache[key] = self.decoder.make_batch(list(key), off[0], off[1])
        _lib.check(gb._lib.mi_gbatch_set_offsets(gb._h, int(off[0]), int(off[1])), "mi_gbatch_set_offsets")
        return gb

    def add_noise(self, batch, timestep: int, noise=None, seed=None):
        """pl_module.py:55-69: t = linspace(T_max, 1/N, N)[timestep] for every crystal, noisy = corruption.sample_marginal(batch, t).
        Returns the reference's triple (noisy_batch, batch, t); the noisy batch carries what the loss needs (`aux`).
        `noise` = (pos [N,3] normal, cell [B,3,3] normal, types [N] uniform) injects the draws (parity tests); default Philox."""
        lib = _lib.load()
        dev = self.device
        N_grid = 1000
        t_val = torch.linspace(self.T, 1.0 / N_grid, N_grid)[int(timestep)]
        gb = self._batch_for(batch.num_atoms)
        B, N = gb.num_graphs, gb.num_nodes
        t = torch.full((B,), float(t_val), device=dev)
        f = lambda x: x.to(dev, torch.float32).contiguous()
        pos0, cell0 = f(batch.pos), f(batch.cell)
        types0 = batch.atomic_numbers.to(dev, torch.int32).contiguous()
        nz = (None, None, None) if noise is None else tuple(f(x) for x in noise)
        pos, cell, types = torch.empty(N, 3, device=dev), torch.empty(B, 3, 3, device=dev), torch.empty(N, dtype=torch.int32, device=dev)
        delta, eps, masked = torch.empty(N, 3, device=dev), torch.empty(B, 3, 3, device=dev), torch.empty(N, dtype=torch.int32, device=dev)
        self._noise_calls = getattr(self, "_noise_calls", 0) + 1
        seed = getattr(self, "noise_seed", 0) if seed is None else seed
        corr = self._corr()
        _lib.check(lib.mi_mg_sample_marginal(gb._h, C.byref(corr), _ptr(pos0), _ptr(cell0), _ptr(types0), _ptr(t), seed, self._noise_calls & 0xFFFFFFFF,
                                             _ptr(nz[0]), _ptr(nz[1]), _ptr(nz[2]), _ptr(pos), _ptr(cell), _ptr(types), _ptr(delta), _ptr(eps), _ptr(masked),
                                             _stream()), "mi_mg_sample_marginal")
        na = gb.num_atoms
        std = (self.corruption["sigma_min"] ** (1 - t) * self.corruption["sigma_max"] ** t * na.float() ** (-1.0 / 3.0))[gb.batch][:, None]
        tau = torch.clamp(torch.ceil(t * self.corruption["d3pm_steps"] - 1e-6), 1, self.corruption["d3pm_steps"])[gb.batch]
        noisy = dict(pos=pos, cell=cell, atomic_numbers=types.long(), num_atoms=na, aux=dict(delta=delta, std=std, eps=eps, tau=tau, masked=masked.bool()),
                     counts=torch.tensor(gb.num_atoms_list))
        return noisy, batch, t

    def predict(self, noised_input):
        """The model output alone (pl_module.py:73) -- library kernels only, no torch arithmetic: the part of calc_sample_loss that may
        run on a side stream beside another network's forward (finetune._ft_step_module_surface)."""
        noisy, batch, t = noised_input
        gb = self._batch_for(noisy["counts"])   # THIS module's workspace (agent and prior must not share one)
        return self.decoder(noisy["pos"], noisy["cell"], noisy["atomic_numbers"], t, gb)

    def calc_sample_loss(self, noised_input, pred=None):
        """pl_module.py:71-81 + SampleLoss (loss.py:36-78): per-crystal sum_field w_field * loss_field, and the model output
        (`pred`: the output of `predict`, when the caller has run the network already)."""
        noisy, batch, t = noised_input
        gb, aux = self._batch_for(noisy["counts"]), noisy["aux"]
        if pred is None:
            pred = self.predict(noised_input)
        B, n2g = gb.num_graphs, gb.batch
        target = aux["std"] * _d_log_p_wn(aux["delta"], aux["std"])
        l_pos = _scatter_mean(((pred["pos"] - target) ** 2).mean(1), n2g, B)
        l_cell = ((pred["cell"] + aux["eps"]) ** 2).mean(dim=(1, 2))
        logp = torch.log_softmax(pred["atomic_numbers"][:, :NUM_CLASSES - 1], dim=1)
        x0 = batch.atomic_numbers.to(logp.device).long()
        nll = -logp.gather(1, (x0 - 1)[:, None])[:, 0]
        l_types = _scatter_mean(aux["masked"].to(nll.dtype) * nll / aux["tau"] + D3PM_LAMBDA * nll, n2g, B)
        return weighted_field_sum(dict(pos=l_pos, cell=l_cell, atomic_numbers=l_types)), pred

    def calc_kl_reg(self, agent_pred, prior_pred, batch):
        """pl_module.py:83-102."""
        n2g = batch.get_batch_idx("pos").to(agent_pred["pos"].device)
        B = agent_pred["cell"].shape[0]
        k0 = torch.pow(agent_pred["cell"] - prior_pred["cell"].detach(), 2).mean(dim=(1, 2))
        k1 = _scatter_mean(torch.pow(agent_pred["pos"] - prior_pred["pos"].detach(), 2).mean(dim=1), n2g, B)
        k2 = _scatter_mean(torch.pow(agent_pred["atomic_numbers"] - prior_pred["atomic_numbers"].detach(), 2).mean(dim=1), n2g, B)
        return k0 + k1 + k2

    @torch.no_grad()
    def sample(self, num_atoms, n_steps=1000, eps_t=1e-3, seed=0, noise=None, i_stop=None, node_offset=0, graph_offset=0, i_start=0, state=None,
               chains=1):
        """Predictor-corrector reverse chain (what draw_samples_from_sampler drives, sample.py:27-64): returns (sample, mean) dicts
        with pos / cell / atomic_numbers / num_atoms.  `noise` (dict of [n_steps, ...] tensors + init_pos / init_cell) injects the draws.

        `chains` > 1 splits the crystals into that many contiguous groups whose chains run CONCURRENTLY on separate HIP streams, as
        DiffCSPModule.sample does: crystals never interact, the Philox draws are indexed by global atom / crystal id and the corrector's
        step sizes are per crystal, so the result is bit for bit that of the groups sampled one after the other (tested, also at the
        benchmark size of four 64-crystal chains); one chain's graph / triplet / node-level kernels and the tails of its dense layers
        then run under the other chains' dense layers.  Against the unsplit batch the samples agree to the plane format's rounding
        only: the power-of-two scales of the plane sets are derived from batch-wide maxima.
        None = automatic (by the number of atoms; what MatterGenSampler.generate uses)."""
        na_all = [int(v) for v in torch.as_tensor(num_atoms).tolist()]
        if chains is None:
            # Automatic: two chains from 512 atoms up, four from 2048.  (Round 3 switched this off: at the benchmark size four concurrent
            # chains did not reproduce themselves run to run.  Round 4 found the cause -- packed-fp32 VALU instructions give wrong results in
            # lanes 48-63 when another stream's LDS + MFMA kernel shares the SIMD, scripts/force_fwd_repro.hip -- and the library is now
            # built without those instructions, matinvent_amd/build.py: 120 of 120 trials of four concurrent forwards bit-identical.)
            chains = 4 if sum(na_all) >= 2048 else 2 if sum(na_all) >= 512 else 1
        chains = max(1, min(int(chains), len(na_all)))
        if chains > 1 and noise is None:
            import threading
            from .streams import concurrent_streams
            cuts = [len(na_all) * k // chains for k in range(chains + 1)]
            g0 = cuts
            n0 = [sum(na_all[:c]) for c in cuts]
            self.decoder.sync()
            cur = torch.cuda.current_stream()
            # (the library builds a weight block's plane set at its first use after a parameter upload, on the stream of that use; the lazy
            # build is guarded by a mutex and later uses on other streams wait for the block's event -- csrc/gemnet.hip get_wplanes --
            # so the chains can start at once, whichever of them meets a block first)
            ready = cur.record_event()
            pool = concurrent_streams(chains, self.device)
            out, err = [None] * chains, [None] * chains

            def run(k):
                try:
                    with torch.cuda.stream(pool[k]):
                        pool[k].wait_event(ready)
                        st = None if state is None else dict(pos=state["pos"][n0[k]:n0[k + 1]], cell=state["cell"][g0[k]:g0[k + 1]],
                                                            atomic_numbers=state["atomic_numbers"][n0[k]:n0[k + 1]])
                        out[k] = self._sample_chain(na_all[g0[k]:g0[k + 1]], n_steps, eps_t, seed, None, i_stop, node_offset + n0[k], graph_offset + g0[k],
                                                    i_start, st, slot=k + 1)
                        cur.wait_event(pool[k].record_event())
                except BaseException as e:   # re-raised on the caller's thread
                    err[k] = e
            th = [threading.Thread(target=run, args=(k,)) for k in range(chains)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            for e in err:
                if e is not None:
                    raise e
            self._last_chain_batches = [o[2] for o in out]
            merged = []
            for which in (0, 1):
                ds = [o[which] for o in out]
                for d in ds:   # allocated on the chain streams, consumed on the caller's
                    for v in d.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(cur)
                merged.append({k: torch.cat([d[k] for d in ds]) for k in ds[0]})
            return merged[0], merged[1]
        sample, mean, gb = self._sample_chain(na_all, n_steps, eps_t, seed, noise, i_stop, node_offset, graph_offset, i_start, state, slot=0)
        self._last_chain_batches = [gb]
        return sample, mean

    def last_sample_invalid(self):
        """bool [B] (CPU): crystals of the most recent `sample` call whose periodic graph exceeded a capacity at some evaluation (a cell
        that collapsed mid-chain: more than max_neighbors kept pairs of one atom, or more atoms inside the cutoff than the candidate
        lists hold).  The chain takes such a crystal out of the graph and goes on -- no host round trip per evaluation, and one
        diverging crystal does not cost the batch; the caller drops it, as the reference's invalid_filter drops collapsed structures
        one by one (pipeline/filters/opt_filter.py:49-61).  Synchronises the device."""
        torch.cuda.synchronize(self.device)
        flags = [gb.graph_status() for gb in getattr(self, "_last_chain_batches", [])]
        return (torch.cat(flags) != 0) if flags else torch.zeros(0, dtype=torch.bool)

    def _sample_chain(self, num_atoms, n_steps, eps_t, seed, noise, i_stop, node_offset, graph_offset, i_start, state, slot=0):
        """One chain over one batch handle on the current stream."""
        lib = _lib.load()
        dev = self.device
        if slot == 0:
            self.shard_offsets = (node_offset, graph_offset)
            gb = self._batch_for(torch.as_tensor(num_atoms))
        else:
            gb = self._chain_batch(torch.as_tensor(num_atoms), (node_offset, graph_offset), slot)
        B, N = gb.num_graphs, gb.num_nodes
        self.decoder.sync()
        corr = self._corr()
        f = lambda x: x.to(dev, torch.float32).contiguous()
        pos, cell, types = torch.empty(N, 3, device=dev), torch.empty(B, 3, 3, device=dev), torch.empty(N, dtype=torch.int32, device=dev)
        ip = ic = None
        if noise is not None:
            ip, ic = f(noise["init_pos"]), f(noise["init_cell"])
        if state is None:
            _lib.check(lib.mi_mg_sampler_init(gb._h, C.byref(corr), seed, _ptr(ip), _ptr(ic), _ptr(pos), _ptr(cell), _ptr(types), _stream()), "mi_mg_sampler_init")
        else:   # resume at grid point i_start from a given state (noise arrays are then indexed from i_start on)
            # (mi_mg_sampler_run wraps the positions in place before its first step: no torch arithmetic on a chain's stream)
            pos, cell, types = f(state["pos"]).clone(), f(state["cell"]).clone(), state["atomic_numbers"].to(dev, torch.int32).contiguous().clone()
        ts = torch.linspace(self.T, eps_t, n_steps).float().contiguous()
        nzs, keep = None, None
        if noise is not None:
            keep = {k: f(torch.stack(list(noise[k]))) for k in ("corr_pos", "corr_cell", "pred_pos", "pred_cell", "pred_u1", "pred_u2")}
            nzs = _lib.MGSamplerNoise(*(keep[k].data_ptr() for k in ("corr_pos", "corr_cell", "pred_pos", "pred_cell", "pred_u1", "pred_u2")))
        mean_pos, mean_cell = pos.clone(), cell.clone()
        stop = n_steps if i_stop is None else i_stop
        _lib.check(lib.mi_mg_sampler_run(self.decoder._h, gb._h, C.byref(corr), n_steps, i_start, stop, ts.numpy().ctypes.data_as(C.POINTER(C.c_float)), seed,
                                         C.byref(nzs) if nzs is not None else None, _ptr(pos), _ptr(cell), _ptr(types), _ptr(mean_pos), _ptr(mean_cell),
                                         _stream()), "mi_mg_sampler_run")
        del keep
        if i_start >= stop:   # a zero-step resume: no predictor ran, the mean IS the state -- which the library has wrapped in place meanwhile
            mean_pos, mean_cell = pos.clone(), cell.clone()
        sample = dict(pos=pos, cell=cell, atomic_numbers=types.long(), num_atoms=gb.num_atoms)
        mean = dict(pos=mean_pos, cell=mean_cell, atomic_numbers=types.long(), num_atoms=gb.num_atoms)
        return sample, mean, gb


# ---- sampler ------------------------------------------------------------------------------------------------------------------------
@dataclass
class MatterGenSampler:
    """models/mattergen/sample.py:127-303: `generate(model, batch_size, num_batches, **kwargs)` -> (list[ChemGraph], list[structure]);
    1000 predictor-corrector steps, the MEAN batch of the last step is what is returned (sample.py:49-50).  Atom counts: the upstream
    ALEX_MP_20 table is not available offline; the mp_20 prior of the DiffCSP side stands in ([UPSTREAM-UNVERIFIED])."""
    batch_size: Optional[int] = None
    num_batches: Optional[int] = None
    num_atoms_distribution: str = "ALEX_MP_20"
    n_steps: int = 1000
    eps_t: float = 1e-3
    seed: int = 0
    discarded: int = 0                  # crystals dropped so far because their periodic graph went over capacity (collapsed cells)
    max_discard_fraction: float = 0.5   # generate() warns per batch and raises when more than this share of a call's request is gone
    _dropped_call: int = 0

    def generate(self, model: MatterGenModule, batch_size=None, num_batches=None, **kwargs):
        from .dist import all_gather_objects, broadcast_object, shard_range
        from .sampling import ATOM_DIST
        batch_size, num_batches = batch_size or self.batch_size, num_batches or self.num_batches
        assert batch_size is not None and num_batches is not None
        rank, world = int(kwargs.get("rank", 0)), int(kwargs.get("world_size", 1))
        model.eval()
        p = ATOM_DIST["mp_20"]
        counts = np.random.choice(len(p), batch_size * num_batches, p=p)
        if world > 1:
            counts = np.asarray(broadcast_object(counts.tolist(), src=0))
        graphs, strucs = [], []
        for bi in range(num_batches):   # unlike the DiffCSP sampler, every batch's samples are kept (sample.py:41-50)
            na = counts[bi * batch_size:(bi + 1) * batch_size]
            lo, hi = shard_range(len(na), rank, world)
            self.seed += 1
            try:
                _, mean = model.sample(na[lo:hi], n_steps=self.n_steps, eps_t=self.eps_t, seed=self.seed, node_offset=int(np.sum(na[:lo])), graph_offset=lo,
                                       chains=kwargs.get("chains"))
                invalid = model.last_sample_invalid()
            except _lib.MIError as e:
                # (only the synchronising form of the forward -- mi_debug_set_mg_nosync(0) -- refuses a batch: there a crystal over a
                #  graph capacity fails the whole call, MI_ECAPACITY, and nothing of the batch can be kept.  A real allocation failure --
                #  MI_ENOMEM -- and every other error are re-raised: they must not shrink the RL step's list silently.)
                if e.code != _lib.MI_ECAPACITY:
                    raise
                import logging
                logging.getLogger(__name__).warning("MatterGenSampler.generate: batch %d discarded (%s)", bi, e)
                _lib.saturation_events(reset=True)
                self._dropped_call += hi - lo
                continue
            _lib.check_saturation("MatterGenSampler.generate")
            from .structure import check_structures_counts   # geometric validity quantities where the final state lives (K18)
            geom = check_structures_counts(mean["num_atoms"], mean["pos"], mean["cell"]).cpu()
            pos, cell, types, nat = (mean[k].detach().cpu() for k in ("pos", "cell", "atomic_numbers", "num_atoms"))
            lengths, angles = lattices_to_params_shape(cell)
            off = [0] + torch.cumsum(nat, 0).tolist()
            # A crystal whose cell collapsed mid-chain (plausible with an untrained or diverging denoiser and the Langevin cell corrector)
            # overflowed the periodic graph's per-atom capacities; the chain took it out of the graph instead of truncating its
            # neighbour list, and ONLY that crystal is dropped here -- as the reference's invalid_filter does (opt_filter.py:49-61).
            n_inv = int(invalid.sum())
            if n_inv:
                import logging
                self._dropped_call += n_inv
                logging.getLogger(__name__).warning("MatterGenSampler.generate: batch %d: %d of %d crystals dropped (periodic graph over capacity: collapsed cells)",
                                                    bi, n_inv, len(nat))
            for i in range(len(nat)):
                if bool(invalid[i]):
                    continue
                g = ChemGraph(pos[off[i]:off[i + 1]], cell[i:i + 1], types[off[i]:off[i + 1]], int(nat[i]))
                g.geometry = {"max_cell_edge": float(geom[i, 0]), "min_distance": float(geom[i, 1]), "volume": float(geom[i, 2])}
                graphs.append(g)
                strucs.append(SimpleStructure(lengths=lengths[i].tolist(), angles=angles[i].tolist(), species=g.atomic_numbers.tolist(),
                                              frac_coords=g.pos.numpy()))
        dropped, self._dropped_call = self._dropped_call, 0
        self.discarded += dropped
        if world > 1:
            parts = all_gather_objects((graphs, strucs, dropped))
            graphs = [g for pp in parts for g in pp[0]]
            strucs = [s for pp in parts for s in pp[1]]
            dropped = sum(pp[2] for pp in parts)
        asked = batch_size * num_batches
        if dropped > self.max_discard_fraction * asked:   # (every rank sees the same counts: they raise or go on together)
            raise RuntimeError(f"MatterGenSampler.generate: {dropped} of {asked} sampled crystals dropped (periodic graph over capacity: the denoiser "
                               f"collapses its cells); more than max_discard_fraction = {self.max_discard_fraction} of the request")
        return graphs, strucs
