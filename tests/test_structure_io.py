"""CPU tests of the structure records / writers / host-side validity filter."""
import math

import numpy as np
import torch

from matinvent_amd import structure as S
from matinvent_amd.data import CrystalData
from matinvent_amd.filters import invalid_filter
from oracle import diffcsp_oracle as O


def _nacl():
    frac = torch.tensor([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5], [.5, .5, .5], [0, 0, .5], [0, .5, 0], [.5, 0, 0]])
    return CrystalData(frac_coords=frac, atom_types=torch.tensor([11] * 4 + [17] * 4), lengths=torch.tensor([[5.64, 5.64, 5.64]]),
                       angles=torch.tensor([[90.0, 90.0, 90.0]]))


def test_lattice_matrix_matches_parameters():
    g = torch.Generator().manual_seed(0)
    for _ in range(8):
        lengths = (3 + 5 * torch.rand(3, generator=g)).tolist()
        angles = (70 + 40 * torch.rand(3, generator=g)).tolist()
        L = S.lattice_matrix(lengths, angles)
        np.testing.assert_allclose(np.linalg.norm(L, axis=1), lengths, rtol=1e-12)
        cosab = lambda a, b: float(L[a] @ L[b] / (np.linalg.norm(L[a]) * np.linalg.norm(L[b])))
        np.testing.assert_allclose([math.degrees(math.acos(cosab(1, 2))), math.degrees(math.acos(cosab(0, 2))),
                                    math.degrees(math.acos(cosab(0, 1)))], angles, rtol=1e-9)
        # same metric tensor (hence volume) as the reference's lattice_params_to_matrix_torch orientation (utils.py:68-96)
        Lr = O.lattice_params_to_matrix(torch.tensor([lengths]), torch.tensor([angles]))[0].double().numpy()
        np.testing.assert_allclose(L @ L.T, Lr @ Lr.T, rtol=2e-5, atol=2e-5)


def test_composition_formula_density():
    d = _nacl()
    assert S.composition(d.atom_types.tolist()) == {"Cl": 4, "Na": 4}
    assert S.reduced_formula(d.atom_types.tolist()) == "ClNa"
    assert S.reduced_formula([8, 8, 8, 26, 26]) == "Fe2O3"
    assert abs(S.volume([5.64] * 3, [90] * 3) - 5.64 ** 3) < 1e-9
    assert abs(S.density(d.atom_types.tolist(), [5.64] * 3, [90] * 3) - 2.164) < 0.01  # rock salt: 2.16 g/cm^3


def test_writers_roundtrip(tmp_path):
    d = _nacl()
    p = S.write_extxyz([d, d], str(tmp_path / "a.extxyz"), infos=[{"reward": 0.5}, {"reward": 0.25}])
    lines = open(p).read().splitlines()
    assert lines[0] == "8" and lines[10] == "8" and len(lines) == 20
    lat = [float(x) for x in lines[1].split('"')[1].split()]
    np.testing.assert_allclose(np.array(lat).reshape(3, 3), 5.64 * np.eye(3), atol=1e-6)
    assert "Properties=species:S:1:pos:R:3" in lines[1] and "reward=0.5" in lines[1] and 'pbc="T T T"' in lines[1]
    sym, x, y, z = lines[3].split()
    assert sym == "Na" and abs(float(x) - 2.82) < 1e-6 and abs(float(y) - 2.82) < 1e-6 and abs(float(z)) < 1e-6
    c = S.write_cif(d, str(tmp_path / "a.cif"))
    txt = open(c).read()
    assert "_cell_length_a   5.64" in txt and "_chemical_formula_sum   'ClNa'" in txt and txt.count("\n  Na  Na") == 4


def test_invalid_filter_thresholds_and_missing_geometry():
    """The filter thresholds the device-side quantities the sampler attached (here: the oracle's numbers stand in for them);
    a record without them is an error, not a silent host-side recomputation."""
    import pytest
    g = torch.Generator().manual_seed(3)
    good = _nacl()
    bad1 = CrystalData(frac_coords=torch.tensor([[0., 0, 0], [0.01, 0, 0]]), atom_types=torch.tensor([1, 1]), lengths=torch.tensor([[5., 5, 5]]),
                       angles=torch.tensor([[90., 90, 90]]))
    bad2 = CrystalData(frac_coords=torch.tensor([[0., 0, 0]]), atom_types=torch.tensor([1]), lengths=torch.tensor([[30., 5, 5]]),
                       angles=torch.tensor([[90., 90, 90]]))
    recs = [bad1, good, bad2]
    fr = torch.cat([d.frac_coords for d in recs])
    lat = torch.stack([torch.from_numpy(S.lattice_matrix(d.lengths[0].tolist(), d.angles[0].tolist())).float() for d in recs])
    chk = O.structure_check(fr, lat, torch.tensor([d.num_atoms for d in recs]))
    for d, c in zip(recs, chk):
        d.geometry = {"max_cell_edge": float(c[0]), "min_distance": float(c[1]), "volume": float(c[2])}
    assert invalid_filter(recs, return_mask=True).tolist() == [False, True, False]
    kept, _ = invalid_filter(recs, None)
    assert kept == [good]
    with pytest.raises(ValueError, match="geometry"):
        invalid_filter([_nacl()])
