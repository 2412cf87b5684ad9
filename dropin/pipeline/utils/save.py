"""pipeline/utils/save.py of the reference (ase / pymatgen writers) on matinvent_amd.structure."""
import os
import pickle

from matinvent_amd.structure import write_extxyz


def save_samples(data_list, save_dir, filename):
    """Pickled sample records (the reference pickles ase.Atoms; here the CrystalData records themselves)."""
    path = os.path.join(save_dir, filename)
    with open(path, "wb") as f:
        pickle.dump(list(data_list), f)
    return path


def save_structures(structures, save_dir, filename):
    """save.py:32-41: extended-XYZ file of the structures."""
    return write_extxyz(structures, os.path.join(save_dir, filename))
