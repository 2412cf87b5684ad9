"""Validity pre-filter of sampled structures (pipeline/filters/opt_filter.py:49-61 `invalid_filter`).

The reference ANDs three tests: mattergen's `structure_validity` (shortest interatomic distance > 0.5 A and cell volume
> 0.1 A^3 in its evaluator -- [UPSTREAM-UNVERIFIED] defaults), SMACT charge neutrality (`is_smact_valid`, needs the SMACT
element database: not reproduced, every composition passes) and `max(lattice.abc) < 25`.  The geometric quantities come
from the device kernel K18 (`matinvent_amd.structure.check_structures`), attached to each record by the sampler
(`CrystalData.geometry`); a record without them is an error -- there is no host-side re-computation.
"""
import logging

import numpy as np

MAX_CELL_EDGE, MIN_DISTANCE, MIN_VOLUME = 25.0, 0.5, 0.1


_warned = []


def invalid_filter(sample_data, sample_struc=None, return_mask=False):
    if not _warned:  # once per process: the reference's filter is stricter than this one
        _warned.append(True)
        logging.warning("invalid_filter: SMACT charge-neutrality test is BYPASSED (its element database is not available); only the "
                        "geometric tests (cell edge < 25 A, shortest distance > 0.5 A, volume > 0.1 A^3) are applied, so compositions the "
                        "reference would reject pass")
    mask = []
    for d in sample_data:
        g = getattr(d, "geometry", None)
        if g is None:
            raise ValueError("invalid_filter: record carries no device-side geometry (CrystalData.geometry); records must come from "
                             "DiffCSPSampler.generate, which attaches it")
        mask.append(g["max_cell_edge"] < MAX_CELL_EDGE and g["min_distance"] > MIN_DISTANCE and g["volume"] > MIN_VOLUME)
    mask = np.array(mask, dtype=bool)
    if return_mask:
        return mask
    data = [x for x, m in zip(sample_data, mask) if m]
    if sample_struc is None:
        return data, None
    return data, [x for x, m in zip(sample_struc, mask) if m]
