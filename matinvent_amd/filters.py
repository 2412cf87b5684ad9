"""Validity pre-filter of sampled structures (pipeline/filters/opt_filter.py:49-61 `invalid_filter`).

The reference ANDs three tests: mattergen's `structure_validity` (shortest interatomic distance > 0.5 A and cell volume
> 0.1 A^3 in its evaluator -- [UPSTREAM-UNVERIFIED] defaults), SMACT charge neutrality (`is_smact_valid`, needs the SMACT
element database: not reproduced, every composition passes) and `max(lattice.abc) < 25`.  The geometric quantities come
from the device kernel K18 (`matinvent_amd.structure.check_structures`), attached to each record by the sampler; records
without them (e.g. replayed from disk) are measured on the host with the same formulae.
"""
import numpy as np

from .structure import lattice_matrix

MAX_CELL_EDGE, MIN_DISTANCE, MIN_VOLUME = 25.0, 0.5, 0.1


def _geometry_host(d):
    L = lattice_matrix(d.lengths.reshape(-1).tolist(), d.angles.reshape(-1).tolist())
    f = d.frac_coords.double().numpy()
    cells = np.array([(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)], dtype=np.float64) @ L
    diff = (f[None, :, :] - f[:, None, :]) @ L
    d2 = ((diff[:, :, None, :] + cells[None, None, :, :]) ** 2).sum(-1)
    n = f.shape[0]
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    keep = (j > i)[:, :, None] | ((j == i)[:, :, None] & (np.arange(27) != 13)[None, None, :])
    return {"max_cell_edge": float(np.linalg.norm(L, axis=1).max()), "min_distance": float(np.sqrt(d2[keep].min())),
            "volume": float(abs(np.linalg.det(L)))}


def invalid_filter(sample_data, sample_struc=None, return_mask=False):
    mask = []
    for d in sample_data:
        g = getattr(d, "geometry", None) or _geometry_host(d)
        mask.append(g["max_cell_edge"] < MAX_CELL_EDGE and g["min_distance"] > MIN_DISTANCE and g["volume"] > MIN_VOLUME)
    mask = np.array(mask, dtype=bool)
    if return_mask:
        return mask
    data = [x for x, m in zip(sample_data, mask) if m]
    if sample_struc is None:
        return data, None
    return data, [x for x, m in zip(sample_struc, mask) if m]
