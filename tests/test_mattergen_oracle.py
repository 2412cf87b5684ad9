"""CPU checks of oracle/mattergen_oracle.py (the restatement the HIP MatterGen-shaped path is tested against) and of the host glue
around it.  PARITY UNPINNED vs upstream (the arithmetic lives in the un-vendored package mattergen @ 5bb2b397): what can be checked
here is internal consistency -- the graph's defining properties, derivative identities, the adapter arithmetic the reference's own
files fix (time grid, loss weights, anchor penalty), and that the C library exposes the same parameter list."""
import ctypes as C
import math

import numpy as np
import torch

from oracle import mattergen_oracle as M

HP = M.GemNetHParams(**M.TINY)


def _case(na, seed=1, scale=5.0):
    g = torch.Generator().manual_seed(seed)
    na = torch.tensor(na)
    N, B = int(na.sum()), len(na)
    frac = torch.rand(N, 3, generator=g)
    cell = scale * torch.eye(3)[None].repeat(B, 1, 1) + 0.5 * M.symmetric_noise(torch.randn(B, 3, 3, generator=g))
    return na, frac, cell, torch.randint(1, 101, (N,), generator=g), g


def test_graph_defining_properties():
    for na, scale in (([4, 7, 1, 10], 5.0), ([20, 3], 3.0), ([1], 2.0), ([0, 5], 4.0)):
        na, frac, cell, a, g = _case(na, scale=scale)
        gr = M.build_graph(frac, cell, na, HP)
        E, N = len(gr["src"]), int(na.sum())
        assert E > 0 and int(gr["rowptr"][-1]) == E
        # every edge has its reverse, the reverse of the reverse is the edge, and the vectors are opposite
        assert torch.equal(gr["swap"][gr["swap"]], torch.arange(E)) and torch.equal(gr["src"][gr["swap"]], gr["dst"])
        assert torch.equal(gr["img"][gr["swap"]], -gr["img"]) and torch.allclose(gr["V"][gr["swap"]], -gr["V"], atol=1e-5)
        # within the cutoff, unit vectors, sorted by target then (source, image)
        assert float(gr["D"].max()) <= HP.cutoff + 1e-5 and float(gr["D"].min()) > 1e-3
        assert torch.allclose((gr["V"] ** 2).sum(1), torch.ones(E), atol=1e-5)
        assert bool((gr["dst"][1:] >= gr["dst"][:-1]).all())
        # edges never leave their crystal
        assert torch.equal(gr["node2graph"][gr["src"]], gr["node2graph"][gr["dst"]])
        # the "upper" representatives any atom selected number at most max_neighbors
        W, R = 2 * HP.max_images + 1, HP.max_images
        code = (gr["img"][:, 0] + R) * W * W + (gr["img"][:, 1] + R) * W + (gr["img"][:, 2] + R)
        zero = R * W * W + R * W + R
        upper = (gr["src"] < gr["dst"]) | ((gr["src"] == gr["dst"]) & (code < zero))
        assert int(torch.bincount(gr["dst"][upper], minlength=N).max()) <= HP.max_neighbors and int(upper.sum()) * 2 == E


def test_basis_functions():
    d = torch.linspace(0.01, HP.cutoff * 1.2, 50)
    rbf = M.radial_basis(d, HP)
    assert rbf.shape == (50, HP.num_radial) and float(rbf[d >= HP.cutoff].abs().max()) == 0.0 and float(rbf.max()) <= 1.0
    # Y_l^0 against closed forms
    c = torch.linspace(-1, 1, 9)
    Y = M.legendre_sph(c, 4)
    assert torch.allclose(Y[:, 0], torch.full_like(c, 0.5 / math.sqrt(math.pi)))
    assert torch.allclose(Y[:, 2], math.sqrt(5 / (4 * math.pi)) * 0.5 * (3 * c * c - 1), atol=1e-6)
    assert torch.allclose(Y[:, 3], math.sqrt(7 / (4 * math.pi)) * 0.5 * (5 * c ** 3 - 3 * c), atol=1e-6)
    z = M.noise_level_encoding(torch.tensor([0.001, 0.5, 1.0]), 64)
    assert z.shape == (3, 64) and torch.allclose(z[:, :32] ** 2 + z[:, 32:] ** 2, torch.ones(3, 32), atol=1e-5)


def test_wrapped_normal_score_is_the_derivative_of_the_log_density():
    x = torch.linspace(-0.45, 0.45, 19, dtype=torch.float64)
    for sigma in (0.05, 0.2, 0.6):
        logp = lambda v: torch.log(sum(torch.exp(-(v + i) ** 2 / 2 / sigma ** 2) for i in range(-10, 11)))
        h = 1e-6
        fd = (logp(x + h) - logp(x - h)) / (2 * h)
        assert torch.allclose(M.d_log_p_wrapped_normal(x, torch.tensor(sigma, dtype=torch.float64)), fd, rtol=1e-5, atol=1e-6)


def test_corruption_marginals_and_adapter_arithmetic():
    corr = M.Corruption()
    # the reference's time grid (pl_module.py:57-62): linspace(T_max, 1/N, N)[timestep]
    assert M.time_grid(corr, 0) == 1.0 and abs(M.time_grid(corr, 999) - 1e-3) < 1e-9
    na, frac, cell, a, g = _case([4, 7, 1, 10])
    N, B = int(na.sum()), len(na)
    batch = dict(pos=frac, cell=cell, atomic_numbers=a, num_atoms=na)
    # t -> 0: nothing moves; the D3PM chain masks a fraction tau / T of the atoms
    t0 = torch.full((B,), 1e-3)
    noise = dict(pos=torch.randn(N, 3, generator=g), cell=torch.randn(B, 3, 3, generator=g), types=torch.rand(N, generator=g))
    noisy, aux = M.sample_marginal(corr, batch, t0, noise)
    d = (noisy["pos"] - frac).abs()
    assert float(torch.minimum(d, 1 - d).max()) < 0.05 and float((noisy["cell"] - cell).abs().max()) < 0.5
    assert float(aux["tau"].max()) == 1.0
    noisy1, aux1 = M.sample_marginal(corr, batch, torch.ones(B), noise)
    assert bool((noisy1["atomic_numbers"] == M.MASK).all())            # t = T: everything is absorbed
    assert torch.allclose(aux1["eps"], aux1["eps"].transpose(1, 2))    # symmetric cell noise
    # unit variance per entry of the symmetric noise
    G = torch.randn(20000, 3, 3, generator=g)
    S = M.symmetric_noise(G)
    assert abs(float(S[:, 0, 1].var()) - 1.0) < 0.05 and abs(float(S[:, 1, 1].var()) - 1.0) < 0.05
    # SampleLoss weights (loss.py:22-26) and the D3PM hybrid term: at tau = 1 a masked atom costs (1 + lambda) nll
    P = M.init_params(HP, seed=0, head_scale=0.3)
    pred = M.gemnet_forward(P, HP, noisy["pos"], noisy["cell"], noisy["atomic_numbers"], na, t0)
    loss, parts = M.sample_loss(corr, batch, aux, pred)
    assert torch.allclose(loss, parts["atomic_numbers"] + parts["cell"] + 0.1 * parts["pos"])
    logp = torch.log_softmax(pred["atomic_numbers"][:, :100], 1)
    nll = -logp.gather(1, (a - 1)[:, None])[:, 0]
    n2g = aux["node2graph"]
    want = M._scatter_mean(aux["masked"].float() * nll + 0.01 * nll, n2g, B)
    assert torch.allclose(parts["atomic_numbers"], want, atol=1e-6)
    # anchor penalty (pl_module.py:83-102): zero for identical predictions, the three mean-squared differences otherwise
    assert float(M.calc_kl_reg(pred, pred, n2g, B).abs().max()) == 0.0
    other = {k: v + 0.1 for k, v in pred.items()}
    assert torch.allclose(M.calc_kl_reg(pred, other, n2g, B), torch.full((B,), 0.03), atol=1e-6)


def test_denoiser_invariances():
    """Translating every atom by a lattice-periodic shift or permuting crystals leaves the per-crystal outputs unchanged: the graph
    is periodic and crystals never interact (what makes the batch shard-able)."""
    P = M.init_params(HP, seed=3, head_scale=0.5)
    na, frac, cell, a, g = _case([5, 8, 3], seed=4)
    t = torch.tensor([0.3, 0.6, 0.9])
    out = M.gemnet_forward(P, HP, frac, cell, a, na, t)
    out_shift = M.gemnet_forward(P, HP, (frac + torch.tensor([0.25, 0.5, 0.125])) % 1.0, cell, a, na, t)
    for k in out:
        assert torch.allclose(out[k], out_shift[k], atol=2e-4 * max(1.0, float(out[k].abs().max()))), k
    # second crystal alone
    sl = slice(5, 13)
    solo = M.gemnet_forward(P, HP, frac[sl], cell[1:2], a[sl], na[1:2], t[1:2])
    assert torch.allclose(solo["pos"], out["pos"][sl], atol=1e-5) and torch.allclose(solo["cell"], out["cell"][1:2], atol=1e-5)
    assert torch.allclose(solo["atomic_numbers"], out["atomic_numbers"][sl], atol=1e-5)


def test_energy_path_and_scale_factors_of_the_parameter_list():
    """The output blocks carry upstream GemNet-T's energy path and ScalingFactors [UPSTREAM-UNVERIFIED]: E_t is returned (the score model never reads
    it), no other output depends on the energy path's tensors, the ScalingFactors are constants (no gradient) that do scale what they stand in front
    of, and param_grads() reports zeros for everything no output depends on."""
    P = M.init_params(HP, seed=3, head_scale=0.5)
    names = [n for n, _, _ in M.param_list(HP)]
    sf = [n for n in names if M.is_scale_factor(n)]
    assert len(sf) == 3 * HP.num_blocks + 2 * (HP.num_blocks + 1) and all(P[n].shape == (1, 1) and float(P[n]) == 1.0 for n in sf)
    na, frac, cell, a, g = _case([5, 8, 3], seed=4)
    t = torch.tensor([0.3, 0.6, 0.9])
    A = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out = M.gemnet_forward(A, HP, frac, cell, a, na, t)
    assert out["energy"].shape == (int(na.sum()), 1) and bool(torch.isfinite(out["energy"]).all())
    gs = dict(zip(A, M.param_grads(out["pos"].square().sum() + out["cell"].square().sum() + out["atomic_numbers"].square().sum(), A, retain_graph=True)))
    for n in names:
        dead = M.is_scale_factor(n) or ".seq_energy." in n or ".out_energy." in n or ".dense_rbf." in n
        assert bool(gs[n].any()) != dead, n
    ge = dict(zip(A, M.param_grads(out["energy"].sum(), A)))
    assert all(bool(ge[f"out_blocks.{i}.out_energy.weight"].any()) for i in range(HP.num_blocks + 1))
    Q = {k: v.clone() for k, v in P.items()}
    Q["out_blocks.0.scale_rbf_F.scale_factor"] = torch.full((1, 1), 1.5)
    Q["int_blocks.0.atom_update.scale_sum.scale_factor"] = torch.full((1, 1), 0.5)
    out2 = M.gemnet_forward(Q, HP, frac, cell, a, na, t)
    assert not torch.allclose(out2["pos"], out["pos"].detach()) and not torch.allclose(out2["atomic_numbers"], out["atomic_numbers"].detach())


def test_library_exposes_the_same_parameter_list():
    """Host-only entry points of the C ABI (no GPU needed): mi_gemnet_create / mi_gemnet_param_info list the oracle's tensors in the
    oracle's order, offsets padded to multiples of 4 floats."""
    from matinvent_amd import _lib
    lib = _lib.load()
    for hpd in (M.TINY, {}):
        hp = M.GemNetHParams(**hpd)
        cfg = _lib.GemNetConfig(hp.emb_atom, hp.emb_edge, hp.emb_trip, hp.emb_rbf, hp.emb_cbf, hp.emb_bil, hp.num_radial, hp.num_spherical, hp.num_blocks,
                                hp.num_before_skip, hp.num_after_skip, hp.num_concat, hp.num_atom, hp.max_neighbors, hp.max_images, hp.cutoff)
        h = C.c_void_p()
        _lib.check(lib.mi_gemnet_create(C.byref(cfg), C.byref(h)))
        want = M.param_list(hp)
        assert lib.mi_gemnet_num_tensors(h) == len(want)
        name, off, numel, r, c = C.c_char_p(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        total = 0
        for i, (n, rows, cols) in enumerate(want):
            _lib.check(lib.mi_gemnet_param_info(h, i, C.byref(name), C.byref(off), C.byref(numel), C.byref(r), C.byref(c)))
            assert (name.value.decode(), r.value, c.value, numel.value) == (n, rows, cols, rows * cols) and off.value == total and total % 4 == 0
            total += (rows * cols + 3) // 4 * 4
        assert lib.mi_gemnet_num_params(h) == total
        lib.mi_gemnet_destroy(h)
    bad = _lib.GemNetConfig(*([64] * 6), 16, 9, 2, 1, 1, 1, 1, 8, 3, 5.0)   # num_spherical = 9
    assert lib.mi_gemnet_create(C.byref(bad), C.byref(h)) == -1 and b"num_spherical" in lib.mi_last_error()


def test_host_records_and_dataset_transform():
    from matinvent_amd.mattergen import ChemGraph, ChemGraphBatch, MatterGenDataset, symmetrize_lattice
    g = torch.Generator().manual_seed(0)
    cells = 4 * torch.eye(3)[None] + torch.randn(3, 3, 3, generator=g)
    S = symmetrize_lattice(cells)
    assert torch.allclose(S, S.transpose(1, 2), atol=1e-5) and torch.allclose(S @ S.transpose(1, 2), cells @ cells.transpose(1, 2), atol=1e-4)
    items = [ChemGraph(torch.rand(n, 3, generator=g), cells[i:i + 1], torch.randint(1, 95, (n,), generator=g)) for i, n in enumerate((3, 5, 2))]
    ds = MatterGenDataset.from_samples(items, np.array([0.1, 0.5, 0.9]))     # dataset.py:42-65
    b = ChemGraphBatch([ds[i] for i in range(3)])
    assert b.get_batch_size() == 3 and b.get_batch_idx("pos").tolist() == [0] * 3 + [1] * 5 + [2] * 2
    assert torch.allclose(b.reward, torch.tensor([0.1, 0.5, 0.9])) and torch.allclose(b.cell, S, atol=1e-6)
    back = b.to_data_list()
    assert [d.num_atoms for d in back] == [3, 5, 2] and torch.equal(back[1].atomic_numbers, items[1].atomic_numbers)


def test_scalar_heads_through_the_derived_radial_tensors_identity():
    """The inference path evaluates the per-edge scalar heads as  F[e] = sum_k rbf[e,k] (x Q^T)[e,k],  Q[k,c] = w[c] Wr[c,k]  (derived at
    mi_gemnet_set_params) instead of  sum_c x[e,c] (rbf Wr^T)[e,c] w[c]  (the oracle's form, oracle/mattergen_oracle.py out block):
    the two are the same contraction in a different order."""
    g = torch.Generator().manual_seed(5)
    E, Ed, Rb = 37, 24, 8
    x, rbf = torch.randn(E, Ed, generator=g, dtype=torch.float64), torch.randn(E, Rb, generator=g, dtype=torch.float64)
    Wr, w = torch.randn(Ed, Rb, generator=g, dtype=torch.float64), torch.randn(Ed, generator=g, dtype=torch.float64)
    ref = (x * (rbf @ Wr.t()) * w).sum(1)
    Q = (w[:, None] * Wr).t()            # [Rb, Ed]
    got = (rbf * (x @ Q.t())).sum(1)
    assert torch.allclose(ref, got, rtol=1e-12, atol=1e-12)


def test_adapter_arithmetic_against_the_reference_generated_fixture():
    """g12_mattergen_adapter.npz was produced by the reference's OWN models/mattergen/pl_module.py and loss.py (imported in the build container through empty
    class stand-ins for the un-vendored `mattergen` package, tests/golden/make_golden.py::g12_mattergen_adapter): the anchor penalty, add_noise's time grid and
    SampleLoss's weighted stack-sum.  The network and the corruptions stay parity-unpinned; this pins what the reference itself computes around them."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_mattergen_adapter.npz"))
    t = lambda k: torch.from_numpy(g[k])
    na = t("kl_num_atoms")
    n2g = torch.repeat_interleave(torch.arange(len(na)), na)
    agent = dict(pos=t("kl_agent_pos"), cell=t("kl_agent_cell"), atomic_numbers=t("kl_agent_types"))
    prior = dict(pos=t("kl_prior_pos"), cell=t("kl_prior_cell"), atomic_numbers=t("kl_prior_types"))
    kl = M.calc_kl_reg(agent, prior, n2g, len(na))
    assert torch.allclose(kl, t("kl_out"), rtol=2e-6, atol=0), (kl - t("kl_out")).abs().max()
    # the time grid: the very float32 values the reference hands its corruption
    for tmax, key in ((1.0, "grid_t_Tmax1p0"), (0.8, "grid_t_Tmax0p8")):
        corr = M.Corruption(T=tmax)
        got = np.array([M.time_grid(corr, int(k)) for k in g["grid_timesteps"]], dtype=np.float32)
        assert np.array_equal(got, g[key]), (got, g[key])
    # SampleLoss: field order, default weights, the hybrid-loss lambda, and the aggregation
    assert list(g["loss_fields"]) == ["pos", "cell", "atomic_numbers"]
    assert [M.WEIGHTS[k] for k in g["loss_fields"]] == list(g["loss_weights"]) and M.D3PM_LAMBDA == float(g["loss_d3pm_hybrid_lambda"])
    fields = dict(pos=t("loss_pos"), cell=t("loss_cell"), atomic_numbers=t("loss_types"))
    assert torch.equal(M.weighted_field_sum(fields), t("loss_agg"))
    wc = dict(zip(g["loss_fields"], g["loss_weights_custom"]))
    assert torch.equal(M.weighted_field_sum(fields, wc), t("loss_agg_custom"))
    assert torch.allclose(torch.stack([v.mean() for v in fields.values()]), t("loss_metric_means"))
