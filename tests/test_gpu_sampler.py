"""GPU parity of the reverse sampler (mi_sampler_run through the host mirror) against the
reference-generated 20-step trajectory (tests/golden/g6_sample.npz) and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import diffcsp_oracle as O
from tests.gpu_util import Box, make_module, params_from_golden, wrap_dist

pytestmark = pytest.mark.gpu


def _golden_module(g):
    P = params_from_golden(g)
    T = int(g["T"])
    m = make_module(64, 2, 8, T, P, sigmas_norm=P["sigma_scheduler.sigmas_norm"])
    # scheduler buffers travel with checkpoints (registered buffers); the time-embedding table is
    # recomputed by the reference on every call and is libm-dependent, so pin the generating host's
    m.load_state_dict({k: v for k, v in P.items() if "scheduler" in k}, strict=False)
    m.time_embedding.freqs.copy_(torch.from_numpy(g["time_freqs"]))
    return m, P, T


def _noise(g, T, N, B):
    z = dict(corr_x=torch.zeros(T + 1, N, 3), pred_l=torch.zeros(T + 1, B, 3, 3), pred_t=torch.zeros(T + 1, N, 100),
             pred_x=torch.zeros(T + 1, N, 3))
    for t in range(T, 1, -1):
        for k in z:
            z[k][t] = torch.from_numpy(g[f"n_{k}_{t}"])
    return z


def test_schedule_buffers_match_reference(golden):
    g = golden("g6_sample")
    m, P, T = _golden_module(g)
    from matinvent_amd.schedules import BetaScheduler, SigmaScheduler
    beta, sig = BetaScheduler(T, "cosine"), SigmaScheduler(T, 0.005, 0.5, sigmas_norm=torch.ones(T + 1))
    for k in ("betas", "alphas", "alphas_cumprod", "sigmas"):  # torch.cos / cumprod on this host vs the generating one
        np.testing.assert_allclose(getattr(beta, k).numpy(), P[f"beta_scheduler.{k}"].numpy(), rtol=2e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(sig.sigmas.numpy(), P["sigma_scheduler.sigmas"].numpy(), rtol=1e-7)
    assert torch.equal(m.beta_scheduler.alphas.cpu(), P["beta_scheduler.alphas"])  # checkpoint load path


def _rel_to_scale(a, b, tol, what):
    a, b = a.detach().cpu().numpy(), np.asarray(b)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:.0e} * max|ref| ({scale:.3g})"


def test_teacher_forced_single_steps(golden):
    """Feed the reference state + noise at every step t, compare state t-1 and the log-probs.
    Tolerances (fp32): wrapped distance <= 1e-5 on fractional coords; lattice and type logits
    <= 1e-5 * max|ref| (the T=20 cosine schedule has c0 ~ 1e2 at t=T, states reach O(1e2));
    log-probs <= 1e-4 relative."""
    g = golden("g6_sample")
    m, P, T = _golden_module(g)
    na = g["num_atoms"]
    B, N = len(na), int(na.sum())
    noise = _noise(g, T, N, B)
    box = Box(na)
    for t in range(T, 0, -1):
        init = tuple(torch.from_numpy(g[f"traj_{t}_{k}"]) for k in ("frac_coords", "lattices", "atom_types"))
        final, traj = m.sample(box, step_lr=float(g["step_lr"]), noise=noise, init=init, record=True, t_start=t, t_stop=t - 1)
        assert wrap_dist(final["frac_coords"].cpu().numpy(), g[f"traj_{t-1}_frac_coords"]).max() < 1e-5, t
        _rel_to_scale(final["lattices"], g[f"traj_{t-1}_lattices"], 1e-5, f"lattices t={t}")
        _rel_to_scale(final["atom_types"], g[f"traj_{t-1}_atom_types"], 1e-5, f"atom_types t={t}")
        if t > 1:
            assert wrap_dist(traj[t]["frac_coords_mid"].cpu().numpy(), g[f"traj_{t}_frac_coords_mid"]).max() < 1e-5, t
            for k in ("log_prob_l", "log_prob_t", "log_prob_x"):
                np.testing.assert_allclose(traj[t][k].cpu().numpy(), g[f"traj_{t}_{k}"], rtol=1e-4, atol=1e-4, err_msg=f"{t} {k}")


def test_free_running_chain(golden):
    """Whole 20-step chain with injected noise: round-off compounds -> looser bound."""
    g = golden("g6_sample")
    m, P, T = _golden_module(g)
    na = g["num_atoms"]
    B, N = len(na), int(na.sum())
    init = (torch.from_numpy(g["x_T"]), torch.from_numpy(g["l_T"]), torch.from_numpy(g["t_T"]))
    final, traj = m.sample(Box(na), step_lr=float(g["step_lr"]), noise=_noise(g, T, N, B), init=init, record=True)
    for t in range(T, -1, -1):
        assert wrap_dist(traj[t]["frac_coords"].cpu().numpy(), g[f"traj_{t}_frac_coords"]).max() < 3e-4, t
        np.testing.assert_allclose(traj[t]["lattices"].cpu().numpy(), g[f"traj_{t}_lattices"], rtol=3e-4, atol=3e-4)
        np.testing.assert_allclose(traj[t]["atom_types"].cpu().numpy(), g[f"traj_{t}_atom_types"], rtol=3e-4, atol=3e-4)
        if t > 1:
            for k in ("log_prob_l", "log_prob_t", "log_prob_x"):
                np.testing.assert_allclose(traj[t][k].cpu().numpy(), g[f"traj_{t}_{k}"], rtol=3e-3, atol=3e-3, err_msg=f"{t} {k}")
    # host glue of DiffCSPSampler.generate: argmax + 1 atom types (sample.py:182)
    assert (final["atom_types"].argmax(-1).cpu().numpy() == g["traj_0_atom_types"].argmax(-1)).all()


@pytest.mark.parametrize("H", [64, 192], ids=["H64", "H192-not-a-power-of-two"])
def test_philox_chain_vs_oracle_and_sharding(H):
    """Built-in counter-based noise: the oracle consumes the same stream (restated in numpy).
    Also: a 2-way crystal shard with global offsets reproduces the single-batch samples.  (H = 192: the sampler's heads and chain at a width
    that is a multiple of 64 but not a power of two.)"""
    T, seed = 12, 4242
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=2, num_freqs=8)
    P = O.init_params(hp, seed=1, head_scale=0.1)
    Pd = {k: v for k, v in P.items()}
    torch.manual_seed(99)
    m = make_module(H, 2, 8, T, Pd, sigmas_norm=None)
    sn = torch.cat([torch.ones(1), 0.5 + torch.rand(T)])
    m.sigma_scheduler.sigmas_norm.copy_(sn)
    sch = O.Schedules.make(T, sigmas_norm=sn)
    na = torch.tensor([4, 7, 2, 5])
    noise = O.philox_sampler_noise(seed, na, T)
    of, otraj = O.sample(P, hp, sch, na, noise, step_lr=5e-6)
    final, traj = m.sample(Box(na), step_lr=5e-6, seed=seed, record=True)
    np.testing.assert_allclose(traj[T]["lattices"].cpu().numpy(), noise["l_T"].numpy(), atol=5e-6)
    assert wrap_dist(final["frac_coords"].cpu().numpy(), of["frac_coords"].numpy()).max() < 3e-4
    np.testing.assert_allclose(final["lattices"].cpu().numpy(), of["lattices"].numpy(), rtol=3e-4, atol=3e-4)
    np.testing.assert_allclose(final["atom_types"].cpu().numpy(), of["atom_types"].numpy(), rtol=3e-4, atol=3e-4)
    for t in range(T, 1, -1):
        np.testing.assert_allclose(traj[t]["log_prob_x"].cpu().numpy(), otraj[t]["log_prob_x"].numpy(), rtol=3e-3, atol=3e-3)
    # shard [0:2] and [2:4] with global node/graph offsets
    f0, _ = m.sample(Box(na[:2]), step_lr=5e-6, seed=seed, node_offset=0, graph_offset=0)
    f1, _ = m.sample(Box(na[2:]), step_lr=5e-6, seed=seed, node_offset=int(na[:2].sum()), graph_offset=2)
    n0 = int(na[:2].sum())
    assert wrap_dist(torch.cat([f0["frac_coords"], f1["frac_coords"]]).cpu().numpy(), final["frac_coords"].cpu().numpy()).max() < 1e-4
    np.testing.assert_allclose(torch.cat([f0["lattices"], f1["lattices"]]).cpu().numpy(), final["lattices"].cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(torch.cat([f0["atom_types"], f1["atom_types"]]).cpu().numpy(), final["atom_types"].cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert n0 == f0["frac_coords"].shape[0]


def test_full_size_chain_properties():
    """BASELINE config-2 shape (B=256 x 20 atoms, H=512, L=6, F=128, T=1000 schedule), a 3-step slice of the
    chain on the default arithmetic path: (1) bit-reproducible for a fixed seed, different for another seed;
    (2) a 2-way crystal shard with global offsets reproduces the single-batch states to fp32 round-off;
    (3) states stay finite, coordinates stay in [0, 1]."""
    import numpy as np
    from matinvent_amd.diffcsp import DiffCSPModule
    torch.manual_seed(0)
    sn = torch.cat([torch.ones(1), torch.linspace(1.2, 0.4, 1000)])
    m = DiffCSPModule(decoder=dict(hidden_dim=512, num_layers=6, num_freqs=128, ln=True, edge_style="fc"),
                      beta_scheduler=dict(timesteps=1000, scheduler_mode="cosine"),
                      sigma_scheduler=dict(timesteps=1000, sigma_begin=0.005, sigma_end=0.5, sigmas_norm=sn), device="cuda")
    with torch.no_grad():
        v = m.decoder.views()
        for k in ("coord_out.weight", "lattice_out.weight", "type_out.weight", "type_out.bias"):
            v[k].mul_(0.01)
    m.decoder.mark_dirty()
    B, n = 256, 20
    full = Box([n] * B)
    a, _ = m.sample(full, step_lr=5e-6, seed=11, t_stop=997)
    a = {k: v.clone() for k, v in a.items() if k in ("frac_coords", "lattices", "atom_types")}
    b, _ = m.sample(full, step_lr=5e-6, seed=11, t_stop=997)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k} not reproducible"
        assert torch.isfinite(a[k]).all()
    assert float(a["frac_coords"].min()) >= 0.0 and float(a["frac_coords"].max()) <= 1.0
    c, _ = m.sample(full, step_lr=5e-6, seed=12, t_stop=997)
    assert not torch.equal(a["lattices"], c["lattices"])
    h = B // 2
    s0, _ = m.sample(Box([n] * h), step_lr=5e-6, seed=11, t_stop=997, node_offset=0, graph_offset=0)
    s1, _ = m.sample(Box([n] * h), step_lr=5e-6, seed=11, t_stop=997, node_offset=h * n, graph_offset=h)
    for k in a:
        cat = torch.cat([s0[k], s1[k]]).cpu().numpy()
        ref = a[k].cpu().numpy()
        if k == "frac_coords":
            assert wrap_dist(cat, ref).max() < 1e-4
        else:
            np.testing.assert_allclose(cat, ref, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(ref).max())))


def test_benchmark_size_concurrent_chains_are_bit_reproducible():
    """The headline configuration -- B = 256 x 20 atoms, H = 512, L = 6, F = 128, four concurrent 64-crystal chains on four streams -- eight
    runs of a 12-step slice: every run gives the bits of the first, and the bits of the four groups sampled ONE AFTER THE OTHER on one stream
    (same kernels, same rows; crystals never interact).  Round 4 found that kernels of two streams sharing a SIMD can corrupt each other's
    packed-fp32 instructions on this chip (scripts/force_fwd_repro.hip); the library is built without them, and this is the check that the
    four-chain headline is covered."""
    from matinvent_amd.diffcsp import DiffCSPModule
    torch.manual_seed(0)
    sn = torch.cat([torch.ones(1), torch.linspace(1.2, 0.4, 1000)])
    m = DiffCSPModule(decoder=dict(hidden_dim=512, num_layers=6, num_freqs=128, ln=True, edge_style="fc"),
                      beta_scheduler=dict(timesteps=1000, scheduler_mode="cosine"),
                      sigma_scheduler=dict(timesteps=1000, sigma_begin=0.005, sigma_end=0.5, sigmas_norm=sn), device="cuda")
    with torch.no_grad():
        v = m.decoder.views()
        for k in ("coord_out.weight", "lattice_out.weight", "type_out.weight", "type_out.bias"):
            v[k].mul_(0.01)
    m.decoder.mark_dirty()
    B, n, keys = 256, 20, ("frac_coords", "lattices", "atom_types")
    kw = dict(step_lr=5e-6, seed=21, t_stop=988)
    first = None
    for run in range(8):
        out, _ = m.sample(Box([n] * B), streams=4, **kw)
        out = {k: out[k].clone() for k in keys}
        assert all(bool(torch.isfinite(out[k]).all()) for k in keys)
        if first is None:
            first = out
        else:
            for k in keys:
                assert torch.equal(out[k], first[k]), f"run {run}: {k} differs from the first run of the same four concurrent chains"
    h = B // 4
    seq = [m.sample(Box([n] * h), streams=1, node_offset=g * h * n, graph_offset=g * h, **kw)[0] for g in range(4)]
    for k in keys:
        assert torch.equal(torch.cat([s[k] for s in seq]), first[k]), f"{k}: four concurrent chains differ from the same groups sampled one after the other"


def test_concurrent_streams_reproduce_the_single_stream_chain():
    """sample(streams=S) splits the crystals into S groups whose chains run concurrently on separate HIP streams; the
    counter-based noise is indexed by global atom / crystal id, so every recorded field equals the unsplit run's (ragged
    groups, S not dividing B, trajectory recording, injected initial state)."""
    T, seed = 10, 99
    hp = O.CSPNetHParams(hidden_dim=64, num_layers=2, num_freqs=8)
    P = O.init_params(hp, seed=2, head_scale=0.1)
    m = make_module(64, 2, 8, T, P, sigmas_norm=None)
    na = torch.tensor([4, 7, 2, 5, 9, 1, 6])
    f1, t1 = m.sample(Box(na), step_lr=5e-6, seed=seed, record=True, streams=1)
    for S in (2, 3):
        fS, tS = m.sample(Box(na), step_lr=5e-6, seed=seed, record=True, streams=S)
        assert sorted(tS) == sorted(t1)
        for t in t1:
            assert sorted(tS[t]) == sorted(t1[t])
            for k in t1[t]:
                a, b = tS[t][k].cpu().numpy(), t1[t][k].cpu().numpy()
                if k == "frac_coords" or k == "frac_coords_mid":
                    assert wrap_dist(a, b).max() < 2e-5, (S, t, k)
                elif a.dtype.kind in "iu":
                    assert (a == b).all(), (S, t, k)
                else:
                    np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4, err_msg=f"{S} {t} {k}")
    # injected initial state is sliced per group
    init = (t1[T]["frac_coords"], t1[T]["lattices"], t1[T]["atom_types"])
    g1, _ = m.sample(Box(na), step_lr=5e-6, seed=seed, init=init, streams=1)
    g2, _ = m.sample(Box(na), step_lr=5e-6, seed=seed, init=init, streams=2)
    assert wrap_dist(g1["frac_coords"].cpu().numpy(), g2["frac_coords"].cpu().numpy()).max() < 2e-5
    np.testing.assert_allclose(g1["lattices"].cpu().numpy(), g2["lattices"].cpu().numpy(), rtol=2e-4, atol=2e-4)
    # the chains of a split call work IN PLACE on one state the call allocates: the caller's `init` is copied, never written, a second call from the
    # same `init` gives the same bits, and the returned state is ONE tensor per field in the batch's order (no per-group pieces to concatenate)
    keep = tuple(v.clone() for v in init)
    g3, _ = m.sample(Box(na), step_lr=5e-6, seed=seed, init=init, streams=2)
    for a, b in zip(init, keep):
        assert torch.equal(a, b)
    for k in ("frac_coords", "lattices", "atom_types"):
        assert torch.equal(g2[k], g3[k]), k
        assert g3[k].is_contiguous() and g3[k].shape[0] == (int(na.sum()) if k != "lattices" else len(na))
    assert g3["frac_coords"].data_ptr() != init[0].data_ptr()
    assert torch.equal(g3["batch_idx"].cpu(), torch.repeat_interleave(torch.arange(len(na)), na)) and torch.equal(g3["num_atoms"].cpu(), na)


def test_benchmark_workload_first_steps_vs_oracle():
    """The benchmark's own workload (bench.py: H=512, L=6, F=128, T=1000, 20-atom crystals, counter-based noise, heads x 1e-2) for
    the first two denoising steps of the chain, one 64-crystal group sampled as two concurrent chains, against the oracle consuming
    the same Philox stream.  (A full 1000-step chain is chaotic in fp32; its properties are checked above.)"""
    import bench
    T, seed = bench.T, bench.SEED_NOISE
    hp = O.CSPNetHParams(hidden_dim=bench.H, num_layers=bench.L, num_freqs=bench.F)
    P = O.init_params(hp, seed=bench.SEED_W, head_scale=bench.HEAD_SCALE)
    m = bench.build_module(torch.device("cuda", 0))
    m.decoder.load_state_dict({k[len("decoder."):]: v for k, v in P.items()})
    na = torch.full((64,), bench.NATOM, dtype=torch.long)
    final, _ = m.sample(Box(na), step_lr=bench.STEP_LR, seed=seed, t_start=T, t_stop=T - 2, streams=2)
    torch.cuda.synchronize()
    sch = O.Schedules.make(T, sigmas_norm=m.sigma_scheduler.sigmas_norm.cpu())
    sch.beta = {k: getattr(m.beta_scheduler, k).cpu() for k in ("betas", "alphas", "alphas_cumprod", "sigmas")}
    noise = O.philox_sampler_noise(seed, na, T, t_stop=T - 2)
    of, _ = O.sample(P, hp, sch, na, noise, step_lr=bench.STEP_LR, t_stop=T - 2, keep_traj=False)
    assert wrap_dist(final["frac_coords"].cpu().numpy(), of["frac_coords"].numpy()).max() < 2e-5
    for k in ("lattices", "atom_types"):
        a, b = final[k].cpu().numpy(), of[k].numpy()
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max()), k


def test_csp_mode_chains_golden(golden):
    """CSP mode (diffusion.py:78-79, 283-287, 308-312, 330, 348-349): with cost_coord = 0 the coordinates, with cost_lattice = 0
    the lattice of the batch are the given ones and never move; the rest of the chain and the log-probs follow the reference-
    generated 10-step trajectories (g10)."""
    g = golden("g10_csp_mode")
    T = int(g["T"])
    P = params_from_golden(g)
    for tag, kw in (("kc", dict(cost_coord=0.0)), ("kl", dict(cost_lattice=0.0))):
        m = make_module(64, 2, 8, T, P, sigmas_norm=P["sigma_scheduler.sigmas_norm"], **kw)
        m.load_state_dict({k: v for k, v in P.items() if "scheduler" in k}, strict=False)
        m.time_embedding.freqs.copy_(torch.from_numpy(g["time_freqs"]))
        assert m.keep_coords == (tag == "kc") and m.keep_lattice == (tag == "kl")
        na = g[f"{tag}_num_atoms"]
        B, N = len(na), int(na.sum())
        z = dict(corr_x=torch.zeros(T + 1, N, 3), pred_l=torch.zeros(T + 1, B, 3, 3), pred_t=torch.zeros(T + 1, N, 100), pred_x=torch.zeros(T + 1, N, 3))
        for t in range(T, 1, -1):
            for k in z:
                z[k][t] = torch.from_numpy(g[f"{tag}_n_{k}_{t}"])
        box = Box(na)
        box.frac_coords, box.lengths, box.angles = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("frac_coords", "lengths", "angles"))
        # the drawn part of the initial state comes from the fixture; the kept part from the batch, as in the reference
        x_T = box.frac_coords if tag == "kc" else torch.from_numpy(g[f"{tag}_x_T"])
        from matinvent_amd.data import lattice_params_to_matrix
        l_T = lattice_params_to_matrix(box.lengths, box.angles) if tag == "kl" else torch.from_numpy(g[f"{tag}_l_T"])
        final, traj = m.sample(box, step_lr=5e-6, noise=z, init=(x_T, l_T, torch.from_numpy(g[f"{tag}_t_T"])), record=True)
        for t in range(T, -1, -1):
            assert wrap_dist(traj[t]["frac_coords"].cpu().numpy(), g[f"{tag}_traj_{t}_frac_coords"]).max() < 3e-4, (tag, t)
            np.testing.assert_allclose(traj[t]["lattices"].cpu().numpy(), g[f"{tag}_traj_{t}_lattices"], rtol=3e-4, atol=3e-4)
            np.testing.assert_allclose(traj[t]["atom_types"].cpu().numpy(), g[f"{tag}_traj_{t}_atom_types"], rtol=3e-4, atol=3e-4)
            if t > 1:
                for k in ("log_prob_l", "log_prob_t", "log_prob_x"):
                    np.testing.assert_allclose(traj[t][k].cpu().numpy(), g[f"{tag}_traj_{t}_{k}"], rtol=3e-3, atol=3e-3, err_msg=f"{tag} {t} {k}")
        if tag == "kc":
            assert wrap_dist(final["frac_coords"].cpu().numpy(), g["kc_frac_coords"] % 1.0).max() == 0
        else:
            np.testing.assert_allclose(final["lattices"].cpu().numpy(), g["kl_traj_0_lattices"], rtol=1e-6, atol=1e-6)
        # without `init`, sample() takes the kept part from the batch itself (diffusion.py:283-287)
        f2, _ = m.sample(box, step_lr=5e-6, seed=3)
        if tag == "kc":
            assert wrap_dist(f2["frac_coords"].cpu().numpy(), g["kc_frac_coords"] % 1.0).max() == 0
        else:
            np.testing.assert_allclose(f2["lattices"].cpu().numpy(), g["kl_traj_0_lattices"], rtol=1e-6, atol=1e-6)


def test_chain_enqueue_does_not_synchronise_once_the_coefficients_are_resident():
    """mi_sampler_run uploads its per-step scalar table only when it differs from the copy the batch holds: the second chain on the
    same batch must be enqueued while a long-running kernel still occupies the stream (no hipStreamSynchronize inside)."""
    import time
    from matinvent_amd import _lib
    m = make_module(64, 2, 8, 6, None)
    box = Box([3, 4])
    m.sample(box, step_lr=5e-6, seed=1)
    torch.cuda.synchronize()
    lib = _lib.load()
    _lib.check(lib.mi_debug_spin(int(2.0e9), None))  # ~1 s of busy-wait on the null stream ahead of the chain
    t0 = time.perf_counter()
    m.sample(box, step_lr=5e-6, seed=2)
    t_enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    assert t_total > 0.3 and t_enqueue < 0.5 * t_total, (t_enqueue, t_total)


@pytest.mark.parametrize("H,streams", [(512, 1), (128, 2)])
def test_what_the_predictor_evaluation_keeps_changes_no_bit(H, streams):
    """The corrector half of a step moves the coordinates only (diffusion.py:320-322: l_t_minus_05 = l_t, t_t_minus_05 = t_t), so the predictor
    evaluation keeps the embedding, the lattice term G of every layer and layer 0's LayerNorm + projections of the corrector evaluation in front of it,
    and the corrector evaluation -- of which only pred_x is read (diffusion.py:326-331) -- computes the coordinate head alone.  With all of that switched
    off (mi_debug_set_eval_reuse(0): everything evaluated every time) the recorded chain is the same, bit for bit."""
    from matinvent_amd import _lib
    lib = _lib.load()
    T, seed = 6, 17
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=3, num_freqs=16)
    P = O.init_params(hp, seed=4, head_scale=0.1)
    m = make_module(H, 3, 16, T, P, sigmas_norm=None)
    na = torch.tensor([20] * 40 + [7, 3, 12]) if H == 512 else torch.tensor([5, 9, 2, 7, 11, 4])   # (H = 512: large enough for the register-tile edge GEMMs and the pair mode)
    was = lib.mi_debug_set_eval_reuse(0)
    try:
        f0, t0 = m.sample(Box(na), step_lr=5e-6, seed=seed, record=True, streams=streams)
        assert lib.mi_debug_set_eval_reuse(7) == 0
        f1, t1 = m.sample(Box(na), step_lr=5e-6, seed=seed, record=True, streams=streams)
    finally:
        lib.mi_debug_set_eval_reuse(was)
    assert sorted(t0) == sorted(t1)
    for t in t0:
        for k in t0[t]:
            assert torch.equal(t0[t][k], t1[t][k]), (t, k)
