"""DiffCSPSampler: host glue around DiffCSPModule.sample (models/diffcsp/sample.py:117-201)."""
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from .data import CrystalData, data2struc, lattices_to_params_shape

# atom-count prior of MP-20 (data table, sample.py:42-62) and the generation step size (:82)
ATOM_DIST = {"mp_20": [0.0, 0.0021742334905660377, 0.021079009433962265, 0.019826061320754717, 0.15271226415094338,
                       0.047132959905660375, 0.08464770047169812, 0.021079009433962265, 0.07808814858490566,
                       0.03434551886792453, 0.0972877358490566, 0.013303360849056603, 0.09669811320754718,
                       0.02155807783018868, 0.06522700471698113, 0.014372051886792452, 0.06703272405660378,
                       0.00972877358490566, 0.053176591981132074, 0.010576356132075472, 0.08995430424528301]}
DEFAULT_STEP_LR = {"gen": {"perov_5": 1e-6, "carbon_24": 1e-5, "mp_20": 5e-6}}


class SampleDataset:
    """sample.py:117-138: draws num_atoms from the numpy GLOBAL generator, like the reference."""

    def __init__(self, total_num, dataset="mp_20"):
        self.total_num = total_num
        self.distribution = ATOM_DIST[dataset]
        self.num_atoms = np.random.choice(len(self.distribution), total_num, p=self.distribution)

    def __len__(self):
        return self.total_num


class _AtomCounts:
    def __init__(self, num_atoms):
        self.num_atoms = torch.as_tensor(np.asarray(num_atoms), dtype=torch.long)


@dataclass
class DiffCSPSampler:
    batch_size: Optional[int] = None
    num_batches: Optional[int] = None
    target_compositions_dict: Optional[list] = None
    num_atoms_distribution: str = "mp_20"
    seed: int = 0

    def generate(self, model, batch_size=None, num_batches=None, **kwargs) -> Tuple[List[CrystalData], list]:
        """sample.py:148-201.  Extra kwargs (`max_num`, `filter`, ...) are tolerated like the reference.
        As in the reference, every batch is sampled but only the LAST batch's outputs are unpacked
        (sample.py:166-177).  `rank`/`world_size` kwargs shard the batch by crystal (DP): each rank
        samples a contiguous block with global noise offsets and the records are all-gathered."""
        batch_size = batch_size or self.batch_size
        num_batches = num_batches or self.num_batches
        assert batch_size is not None and num_batches is not None
        rank, world = int(kwargs.get("rank", 0)), int(kwargs.get("world_size", 1))
        model.eval()
        dataset = SampleDataset(total_num=batch_size * num_batches, dataset=self.num_atoms_distribution)
        from .dist import collectives_on
        if world > 1 or collectives_on():
            # the atom counts come from numpy's unseeded GLOBAL generator (sample.py:123): every rank would draw a different
            # vector, while the shard ranges and the global noise offsets below assume ONE.  Rank 0's draw is the batch.
            from .dist import broadcast_object
            dataset.num_atoms = np.asarray(broadcast_object(dataset.num_atoms.tolist(), src=0))
        step_lr = DEFAULT_STEP_LR["gen"]["mp_20"]
        outputs = None
        for bi in range(num_batches):
            na = dataset.num_atoms[bi * batch_size:(bi + 1) * batch_size]
            from .dist import shard_range
            lo, hi = shard_range(len(na), rank, world)
            node_off = int(np.sum(na[:lo]))
            self.seed += 1
            counts = _AtomCounts(na[lo:hi])
            outputs, _ = model.sample(counts, step_lr=step_lr, seed=self.seed, node_offset=node_off, graph_offset=lo)
        from . import _lib
        if hasattr(model, "check_graph"):
            model.check_graph()   # (knn edge style: the chains' neighbour lists stayed inside their capacity -- the device is about to be drained anyway)
        _lib.check_saturation("DiffCSPSampler.generate")  # (the results are about to be copied to the host: the device is drained anyway)
        # geometric validity quantities of the final state, computed where it lives (K18); the filter step thresholds them
        from .structure import check_structures
        geom = check_structures(model.crystal_batch(counts, node_off, lo), outputs["frac_coords"], outputs["lattices"]).cpu()
        frac_coords = outputs["frac_coords"].detach().cpu()
        num_atoms = outputs["num_atoms"].detach().cpu()
        atom_types = outputs["atom_types"].detach().cpu()
        lattices = outputs["lattices"].detach().cpu()
        lengths, angles = lattices_to_params_shape(lattices)
        atom_types = torch.argmax(atom_types, dim=-1) + 1  # sample.py:182
        offset = [0] + torch.cumsum(num_atoms, dim=0).tolist()
        data_list, struc_list = [], []
        for i in range(len(num_atoms)):
            d = CrystalData(frac_coords=frac_coords[offset[i]:offset[i + 1]], atom_types=atom_types[offset[i]:offset[i + 1]],
                            lengths=lengths[i].view(1, -1), angles=angles[i].view(1, -1), num_atoms=int(num_atoms[i]))
            d.geometry = {"max_cell_edge": float(geom[i, 0]), "min_distance": float(geom[i, 1]), "volume": float(geom[i, 2])}
            data_list.append(d)
            struc_list.append(data2struc(d))
        if world > 1 or collectives_on():
            from .dist import all_gather_objects
            parts = all_gather_objects((data_list, struc_list))
            data_list = [d for p in parts for d in p[0]]
            struc_list = [s for p in parts for s in p[1]]
        return data_list, struc_list
