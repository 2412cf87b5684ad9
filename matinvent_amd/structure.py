"""Structure records, geometric pre-filter and writers.

What the reference gets from pymatgen / ase on the sampler -> filter -> reward hand-off, without either package:
  * `check_structures`: per-crystal longest cell edge, shortest interatomic distance (27 images) and volume computed on
    the device from the sampler's final state (K18, `mi_structure_check`), and `geometric_mask`, the cell test of
    pipeline/filters/opt_filter.py:53-55 (`max(abc) < 25`) plus the distance / volume thresholds of the external
    `structure_validity` check (0.5 A, 0.1 A^3 -- [UPSTREAM-UNVERIFIED] defaults of mattergen's evaluator).  The charge-
    neutrality test (SMACT) of `invalid_filter` needs its element database and is not reproduced.
  * `lattice_matrix` (pymatgen `Lattice.from_parameters` orientation), `volume`, `density`, `composition`,
    `reduced_formula` -- the keys memory/replay_buffer.py:38 and memory/ltm.py:31 dedupe on;
  * `write_extxyz` / `write_cif` -- pipeline/utils/save.py:32-41 writes extxyz through ase.
"""
import ctypes as C
import math
from functools import reduce
from typing import Iterable, List

import numpy as np
import torch

SYMBOLS = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc "
           "Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi "
           "Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm").split()
# standard atomic weights (u), Z = 1..100
MASSES = (0.0, 1.008, 4.0026, 6.94, 9.0122, 10.81, 12.011, 14.007, 15.999, 18.998, 20.180, 22.990, 24.305, 26.982, 28.085, 30.974,
          32.06, 35.45, 39.948, 39.098, 40.078, 44.956, 47.867, 50.942, 51.996, 54.938, 55.845, 58.933, 58.693, 63.546, 65.38,
          69.723, 72.630, 74.922, 78.971, 79.904, 83.798, 85.468, 87.62, 88.906, 91.224, 92.906, 95.95, 98.0, 101.07, 102.91,
          106.42, 107.87, 112.41, 114.82, 118.71, 121.76, 127.60, 126.90, 131.29, 132.91, 137.33, 138.91, 140.12, 140.91, 144.24,
          145.0, 150.36, 151.96, 157.25, 158.93, 162.50, 164.93, 167.26, 168.93, 173.05, 174.97, 178.49, 180.95, 183.84, 186.21,
          190.23, 192.22, 195.08, 196.97, 200.59, 204.38, 207.2, 208.98, 209.0, 210.0, 222.0, 223.0, 226.0, 227.0, 232.04, 231.04,
          238.03, 237.0, 244.0, 243.0, 247.0, 247.0, 251.0, 252.0, 257.0)
assert len(SYMBOLS) == 101 and len(MASSES) == 101
_AMU_PER_A3_TO_G_CM3 = 1.66053906660


def lattice_matrix(lengths, angles_deg) -> np.ndarray:
    """pymatgen `Lattice.from_parameters(a, b, c, alpha, beta, gamma)`: c along z, a in the xz plane."""
    a, b, c = (float(x) for x in lengths)
    al, be, ga = (math.radians(float(x)) for x in angles_deg)
    val = (math.cos(al) * math.cos(be) - math.cos(ga)) / (math.sin(al) * math.sin(be))
    val = max(-1.0, min(1.0, val))
    gs = math.acos(val)
    va = [a * math.sin(be), 0.0, a * math.cos(be)]
    vb = [-b * math.sin(al) * math.cos(gs), b * math.sin(al) * math.sin(gs), b * math.cos(al)]
    return np.array([va, vb, [0.0, 0.0, c]], dtype=np.float64)


def composition(species: Iterable[int]) -> dict:
    out = {}
    for z in species:
        s = SYMBOLS[int(z)]
        out[s] = out.get(s, 0) + 1
    return dict(sorted(out.items()))


def reduced_formula(species: Iterable[int]) -> str:
    """Counts divided by their gcd, elements in alphabetical order (a canonical dedupe key; pymatgen orders by
    electronegativity, which only changes the spelling)."""
    comp = composition(species)
    if not comp:
        return ""
    g = reduce(math.gcd, comp.values())
    return "".join(f"{k}{v // g if v // g > 1 else ''}" for k, v in comp.items())


def volume(lengths, angles_deg) -> float:
    return float(abs(np.linalg.det(lattice_matrix(lengths, angles_deg))))


def density(species, lengths, angles_deg) -> float:
    """g / cm^3."""
    m = sum(MASSES[int(z)] for z in species)
    return m * _AMU_PER_A3_TO_G_CM3 / volume(lengths, angles_deg)


def _fields(s):
    """(lengths, angles, species, frac) of a SimpleStructure / CrystalData-like record."""
    if hasattr(s, "species"):
        return list(s.lengths), list(s.angles), [int(z) for z in s.species], np.asarray(s.frac_coords, dtype=np.float64)
    return (s.lengths.reshape(-1).tolist(), s.angles.reshape(-1).tolist(), [int(z) for z in s.atom_types.tolist()],
            s.frac_coords.double().numpy())


def write_extxyz(structures, path: str, infos: List[dict] = None) -> str:
    """Extended XYZ as ase writes it for periodic Atoms: `Lattice="..." Properties=species:S:1:pos:R:3 pbc="T T T"`."""
    with open(path, "w") as f:
        for k, s in enumerate(structures):
            lengths, angles, species, frac = _fields(s)
            L = lattice_matrix(lengths, angles)
            cart = frac @ L
            extra = "".join(f" {a}={b}" for a, b in (infos[k] if infos else {}).items())
            f.write(f"{len(species)}\n")
            f.write('Lattice="' + " ".join(f"{x:.8f}" for x in L.reshape(-1)) + '" Properties=species:S:1:pos:R:3' + extra + ' pbc="T T T"\n')
            for z, p in zip(species, cart):
                f.write(f"{SYMBOLS[z]:<2s} {p[0]:16.8f} {p[1]:16.8f} {p[2]:16.8f}\n")
    return path


def write_cif(structure, path: str, name: str = "generated") -> str:
    """Minimal P1 CIF (cell parameters + fractional sites)."""
    lengths, angles, species, frac = _fields(structure)
    lines = [f"data_{name}", "_symmetry_space_group_name_H-M   'P 1'", "_symmetry_Int_Tables_number   1",
             f"_cell_length_a   {lengths[0]:.6f}", f"_cell_length_b   {lengths[1]:.6f}", f"_cell_length_c   {lengths[2]:.6f}",
             f"_cell_angle_alpha   {angles[0]:.6f}", f"_cell_angle_beta   {angles[1]:.6f}", f"_cell_angle_gamma   {angles[2]:.6f}",
             f"_chemical_formula_sum   '{reduced_formula(species)}'", f"_cell_volume   {volume(lengths, angles):.6f}", "loop_",
             " _atom_site_type_symbol", " _atom_site_label", " _atom_site_fract_x", " _atom_site_fract_y", " _atom_site_fract_z",
             " _atom_site_occupancy"]
    for k, (z, p) in enumerate(zip(species, frac)):
        lines.append(f"  {SYMBOLS[z]}  {SYMBOLS[z]}{k}  {p[0] % 1.0:.8f}  {p[1] % 1.0:.8f}  {p[2] % 1.0:.8f}  1")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


# ---- device-side geometric pre-filter (K18) ---------------------------------------------------------------------------
def check_structures(batch, frac_coords: torch.Tensor, lattices: torch.Tensor) -> torch.Tensor:
    """[B,4] on the device: longest cell edge, shortest interatomic distance (27 images), volume, atom count."""
    from . import _lib
    lib = _lib.load()
    fr = frac_coords.detach().float().contiguous()
    lat = lattices.detach().float().contiguous()
    assert fr.is_cuda and lat.is_cuda
    out = torch.empty(batch.num_graphs, 4, device=fr.device)
    _lib.check(lib.mi_structure_check(batch._h, C.c_void_p(fr.data_ptr()), C.c_void_p(lat.data_ptr()), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mi_structure_check")
    return out


def check_structures_counts(num_atoms: torch.Tensor, frac_coords: torch.Tensor, lattices: torch.Tensor) -> torch.Tensor:
    """check_structures for a batch described by its atom counts only (device tensors)."""
    from . import _lib
    lib = _lib.load()
    fr, lat = frac_coords.detach().float().contiguous(), lattices.detach().float().contiguous()
    off = torch.zeros(len(num_atoms) + 1, dtype=torch.int32, device=fr.device)
    off[1:] = torch.cumsum(num_atoms.to(fr.device), 0)
    out = torch.empty(len(num_atoms), 4, device=fr.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.mi_structure_check_offsets(p(off), len(num_atoms), p(fr), p(lat), p(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "mi_structure_check_offsets")
    return out


def geometric_mask(check: torch.Tensor, max_cell: float = 25.0, min_dist: float = 0.5, min_volume: float = 0.1) -> torch.Tensor:
    """max(abc) < 25 (opt_filter.py:53-55) and the distance / volume thresholds of `structure_validity`."""
    return (check[:, 0] < max_cell) & (check[:, 1] > min_dist) & (check[:, 2] > min_volume)
