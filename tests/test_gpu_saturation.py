"""The two-plane fp16 operand format has a finite range (65504 / scale per plane set).  Every conversion that clamps counts
itself on the device (mi_saturation_events); nothing that left the range may come back as silent finite garbage."""
import numpy as np
import pytest
import torch

from oracle import diffcsp_oracle as O
from tests.gpu_util import Box, make_module

pytestmark = pytest.mark.gpu


def _inputs(na, gen, lat_scale):
    B, N = len(na), int(sum(na))
    n2g = torch.repeat_interleave(torch.arange(B), torch.tensor(na))
    t_emb = O.time_embedding(torch.full((B,), 9), 256)
    at, fr = torch.randn(N, 100, generator=gen), torch.rand(N, 3, generator=gen)
    lat = lat_scale * torch.randn(B, 3, 3, generator=gen)
    return B, N, n2g, t_emb, at, fr, lat


@pytest.mark.parametrize("lat_scale", [1.0, 1e3, 1e6])
def test_heavy_tailed_states_keep_parity_and_raise_no_flag(lat_scale):
    """Un-scaled random-init heads and lattices up to 1e6 (the Gram term L L^T reaches 1e12, pre-activations 1e11 and more): the
    per-layer activation scales are derived on the device from rigorous bounds, so the format must hold -- parity with the oracle
    at the forward tolerance AND a zero saturation count."""
    from matinvent_amd import _lib
    if _lib.load().mi_plane_format() != 2:
        pytest.skip("three-plane bf16 build: no range limit")
    H, L, F = 128, 3, 16
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=11)
    gen = torch.Generator().manual_seed(5)
    m = make_module(H, L, F, 20, P)
    na = [20] * 40 + [3, 7, 1]   # 16k edges: the plane-GEMM edge path and the node-level plane path
    B, N, n2g, t_emb, at, fr, lat = _inputs(na, gen, lat_scale)
    _lib.saturation_events(reset=True)
    pl, px, pt = m.decoder(t_emb.cuda(), at.cuda(), fr.cuda(), lat.cuda(), torch.tensor(na))
    n_sat = _lib.saturation_events(reset=True)
    ol, ox, ot = O.cspnet_forward(P, hp, t_emb, at, fr, lat, torch.tensor(na), n2g)
    assert n_sat == 0, f"{n_sat} conversions saturated at lattice scale {lat_scale:g}"
    for name, a, b in (("pred_l", pl, ol), ("pred_x", px, ox), ("pred_t", pt, ot)):
        assert bool(torch.isfinite(a).all()), name
        err = float((a.cpu() - b).abs().max()) / max(1.0, float(b.abs().max()))
        assert err <= 5e-5, f"{name}: {err:.3e} of max|ref| at lattice scale {lat_scale:g}"


@pytest.mark.parametrize("what", ["weight", "layernorm", "nan"])
def test_values_beyond_the_format_raise_the_flag(what):
    """The fixed-scale operand classes do have limits (weights beyond 1023, LayerNorm outputs beyond 8188), and NaN / inf upstream
    clamps to finite values: each must be COUNTED, and the host glue must refuse to hand such results on."""
    from matinvent_amd import _lib
    if _lib.load().mi_plane_format() != 2:
        pytest.skip("three-plane bf16 build: no range limit")
    H, L, F = 128, 2, 16
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=12)
    if what == "weight":
        P["decoder.csp_layer_1.edge_mlp.2.weight"][3, 5] = 5000.0
    elif what == "layernorm":
        P["decoder.csp_layer_0.layer_norm.weight"][:] = 4000.0
    gen = torch.Generator().manual_seed(6)
    m = make_module(H, L, F, 20, P)
    na = [20] * 40
    B, N, n2g, t_emb, at, fr, lat = _inputs(na, gen, 1.0)
    if what == "nan":
        at[17, 3] = float("nan")
    _lib.saturation_events(reset=True)
    m.decoder(t_emb.cuda(), at.cuda(), fr.cuda(), lat.cuda(), torch.tensor(na))
    assert _lib.saturation_events(reset=False) > 0
    with pytest.raises(FloatingPointError, match="saturated"):
        _lib.check_saturation("test")
    assert _lib.saturation_events(reset=True) == 0   # the check cleared the counter


def test_clean_chain_reports_zero():
    from matinvent_amd import _lib
    m = make_module(64, 2, 8, 6, None)
    _lib.saturation_events(reset=True)
    m.sample(Box([5, 20, 3]), step_lr=5e-6, seed=3)
    assert _lib.saturation_events(reset=True) == 0


@pytest.mark.parametrize("what", ["nan-weight", "nan-embedding"])
def test_the_mattergen_shaped_network_counts_its_saturations_too(what):
    """Every translation unit that converts to the plane format owns a copy of the device counter (no relocatable device code); each
    registers its reader with mi_saturation_events at library load, so a NaN that reaches a plane conversion of the MatterGen-shaped
    network (csrc/gemnet.hip) -- in a weight block's plane set, or in the activations downstream of a NaN embedding row -- must show up in
    the count, and check_saturation (what MatterGenSampler.generate and ft_step call) must then refuse the result.  (A NaN weight gives
    FINITE outputs: the conversion clamps it, which is exactly the silent garbage the counter exists for.)"""
    from matinvent_amd import _lib
    from matinvent_amd.mattergen import MatterGenModule
    from oracle import mattergen_oracle as MO
    if _lib.load().mi_plane_format() != 2:
        pytest.skip("three-plane bf16 build: no range limit")
    hpd = dict(MO.TINY, emb_atom=128, emb_edge=128, max_neighbors=50, cutoff=7.0)
    hp = MO.GemNetHParams(**hpd)
    g = torch.Generator().manual_seed(4)
    na = torch.tensor([20] * 12)
    N, B = int(na.sum()), len(na)
    frac = torch.rand(N, 3, generator=g)
    cell = 6.0 * torch.eye(3)[None].repeat(B, 1, 1) + 0.3 * MO.symmetric_noise(torch.randn(B, 3, 3, generator=g))
    a, t = torch.randint(1, 101, (N,), generator=g), 0.1 + 0.8 * torch.rand(B, generator=g)
    for poisoned in (False, True):
        P = MO.init_params(hp, seed=2, head_scale=0.5)
        if poisoned:
            P["int_blocks.0.dense_ca.weight" if what == "nan-weight" else "atom_emb.weight"].view(-1)[7] = float("nan")
        m = MatterGenModule(gemnet=hpd)
        m.decoder.load_state_dict(P, strict=True)
        gb = m.decoder.make_batch(na)
        assert gb.graph(frac, cell)["src"].shape[0] >= 4096   # the plane-set layers engage from 4096 edges up
        _lib.saturation_events(reset=True)
        with torch.no_grad():
            m.decoder(frac, cell, a, t, gb)
        n = _lib.saturation_events(reset=False)
        if not poisoned:
            assert n == 0
            continue
        assert n > 0, "a NaN reached the plane format of the MatterGen-shaped network uncounted"
        with pytest.raises(FloatingPointError):
            _lib.check_saturation("test")
