"""Crystal records and batches (host glue).

Replaces the torch_geometric `Data` / `Batch` / `DataLoader` trio used by
models/diffcsp/sample.py:185-199, models/diffcsp/finetune.py:5-18 and
models/suite/diffcsp.py:116-131 with the few attributes the hot path reads.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch


@dataclass
class CrystalData:
    """One crystal, as DiffCSPSampler.generate emits it (sample.py:185-193)."""
    frac_coords: torch.Tensor            # [n, 3]
    atom_types: torch.Tensor             # [n] long, 1..100
    lengths: torch.Tensor                # [1, 3]
    angles: torch.Tensor                 # [1, 3] degrees
    num_atoms: int = 0
    reward: Optional[torch.Tensor] = None  # [1], attached by CrystalDataset (finetune.py:13)

    def __post_init__(self):
        self.num_atoms = int(self.num_atoms) if self.num_atoms else int(self.frac_coords.shape[0])
        self.num_nodes = self.num_atoms


class CrystalBatchData:
    """Collated crystals: contiguous node layout, `batch` = node -> crystal."""

    def __init__(self, items: List[CrystalData]):
        self.num_graphs = len(items)
        self.num_atoms = torch.tensor([d.num_atoms for d in items], dtype=torch.long)
        self.num_nodes = int(self.num_atoms.sum())
        self.batch = torch.repeat_interleave(torch.arange(self.num_graphs), self.num_atoms)
        cat = lambda xs, shape: torch.cat(xs) if xs else torch.zeros(shape)
        self.frac_coords = cat([d.frac_coords.float() for d in items], (0, 3))
        self.atom_types = cat([d.atom_types.long() for d in items], (0,)).long()
        self.lengths = cat([d.lengths.float().view(1, 3) for d in items], (0, 3))
        self.angles = cat([d.angles.float().view(1, 3) for d in items], (0, 3))
        if items and all(d.reward is not None for d in items):
            self.reward = torch.cat([d.reward.float().view(1) for d in items])

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class CrystalDataset:
    """DiffCSPDataset (finetune.py:5-18): attaches rewards[i] to data_list[i]."""

    def __init__(self, data_list, rewards=None):
        self.data_list = list(data_list)
        if rewards is not None:
            rewards = torch.tensor(np.asarray(rewards), dtype=torch.float)
            for i, d in enumerate(self.data_list):
                d.reward = rewards[i].unsqueeze(dim=0)

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, i):
        return self.data_list[i]


class CrystalLoader:
    """Minimal DataLoader: shuffled (torch global RNG, like torch's sampler) mini-batches."""

    def __init__(self, dataset, batch_size, shuffle=True):
        self.dataset, self.batch_size, self.shuffle = dataset, max(1, int(batch_size)), shuffle

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        for s in range(0, n, self.batch_size):
            yield CrystalBatchData([self.dataset[i] for i in order[s:s + self.batch_size]])


def lattice_params_to_matrix(lengths: torch.Tensor, angles: torch.Tensor) -> torch.Tensor:
    """(a, b, c), (alpha, beta, gamma in degrees) -> 3x3 row-vector lattice; the formula of models/diffcsp/utils.py:68-96 (c along z, a in
    the xz plane), used where the sampler is handed a known lattice (CSP mode, diffusion.py:286-287)."""
    ar = torch.deg2rad(angles)
    co, si = torch.cos(ar), torch.sin(ar)
    gs = torch.arccos(torch.clamp((co[:, 0] * co[:, 1] - co[:, 2]) / (si[:, 0] * si[:, 1]), -1.0, 1.0))
    z = torch.zeros_like(lengths[:, 0])
    va = torch.stack([lengths[:, 0] * si[:, 1], z, lengths[:, 0] * co[:, 1]], dim=1)
    vb = torch.stack([-lengths[:, 1] * si[:, 0] * torch.cos(gs), lengths[:, 1] * si[:, 0] * torch.sin(gs), lengths[:, 1] * co[:, 0]], dim=1)
    vc = torch.stack([z, z, lengths[:, 2]], dim=1)
    return torch.stack([va, vb, vc], dim=1)


def lattices_to_params_shape(lattices: torch.Tensor):
    """3x3 lattice matrices -> (lengths, angles in degrees), models/diffcsp/sample.py:103-114."""
    lengths = torch.sqrt(torch.sum(lattices ** 2, dim=-1))
    angles = torch.zeros_like(lengths)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        angles[..., i] = torch.clamp(torch.sum(lattices[..., j, :] * lattices[..., k, :], dim=-1) / (lengths[..., j] * lengths[..., k]),
                                     -1.0, 1.0)
    return lengths, torch.arccos(angles) * 180.0 / np.pi


@dataclass
class SimpleStructure:
    """Stand-in for pymatgen.Structure when pymatgen is absent: what data2struc (sample.py:87-100)
    feeds the constructor (lattice parameters, species, fractional coordinates)."""
    lengths: List[float]
    angles: List[float]
    species: List[int]
    frac_coords: np.ndarray = field(repr=False, default=None)

    @property
    def num_sites(self):
        return len(self.species)


def data2struc(data: CrystalData):
    """sample.py:87-100; a pymatgen Structure when pymatgen is importable, else SimpleStructure."""
    lengths, angles = data.lengths[0].tolist(), data.angles[0].tolist()
    try:
        from pymatgen.core.lattice import Lattice
        from pymatgen.core.structure import Structure
        return Structure(lattice=Lattice.from_parameters(*(lengths + angles)), species=data.atom_types.numpy(),
                         coords=data.frac_coords.numpy(), coords_are_cartesian=False)
    except ImportError:
        return SimpleStructure(lengths, angles, data.atom_types.tolist(), data.frac_coords.numpy())
