"""Pin the CPU oracle (oracle/diffcsp_oracle.py) against golden vectors produced by the
reference's own vendored DiffCSP code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import diffcsp_oracle as O

T = torch.from_numpy


def params(g, prefix="P__"):
    return {k[len(prefix):]: T(np.asarray(g[k])) for k in g.files if k.startswith(prefix)}


def close(a, b, rtol=1e-6, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


TINY = O.CSPNetHParams(hidden_dim=64, num_layers=2, num_freqs=8)


def test_repeat_blocks_kats(golden):
    g = golden("g1_repeat_blocks")
    for i in range(8):
        ci, st, bi, ri = g[f"c{i}_flags"].tolist()
        rep = g[f"c{i}_repeats"].tolist()
        rep = rep[0] if len(rep) == 1 else rep
        out = O.repeat_blocks(g[f"c{i}_sizes"].tolist(), rep, bool(ci), st, bi, ri)
        assert out.tolist() == g[f"c{i}_out"].tolist(), i


def test_schedules(golden):
    g = golden("g2_schedulers")
    for Tn in (20, 1000):
        b = O.beta_tables(Tn, "cosine")
        for k in ("betas", "alphas", "alphas_cumprod", "sigmas"):
            close(b[k], g[f"T{Tn}_beta_{k}"], rtol=0, atol=0)
        s = O.sigma_tables(Tn, 0.005, 0.5, generator=torch.Generator().manual_seed(1234))
        close(s["sigmas"], g[f"T{Tn}_sigma_sigmas"], rtol=0, atol=0)
        # the reference draws sigmas_norm from the *global* generator under manual_seed(1234)
        torch.manual_seed(1234)
        s2 = O.sigma_tables(Tn, 0.005, 0.5)
        close(s2["sigmas_norm"], g[f"T{Tn}_sigma_sigmas_norm"], rtol=1e-6, atol=0)
    for mode in ("linear", "quadratic", "sigmoid"):
        b = O.beta_tables(50, mode)
        close(b["betas"], g[f"T50_{mode}_betas"], rtol=0, atol=0)
        close(b["sigmas"], g[f"T50_{mode}_sigmas"], rtol=0, atol=0)
    close(O.d_log_p_wrapped_normal(T(g["wn_x"]), T(g["wn_sigma"])), g["wn_dlogp"], rtol=1e-6)
    close(O.p_wrapped_normal(T(g["wn_x"]), T(g["wn_sigma"])), g["wn_p"], rtol=1e-6)
    close(O.log_prob_wn(T(g["lp_x"]), T(g["lp_mu"]), T(g["wn_sigma"])), g["lp_out"], rtol=1e-6)


def test_lattice(golden):
    g = golden("g3_lattice")
    m = O.lattice_params_to_matrix(T(g["lengths"]), T(g["angles"]))
    close(m, g["matrix"], rtol=0, atol=0)
    l, a = O.lattices_to_params_shape(m)
    close(l, g["back_lengths"], rtol=0, atol=0)
    close(a, g["back_angles"], rtol=0, atol=0)
    l, a = O.lattices_to_params_shape(T(g["rnd"]))
    close(l, g["rnd_lengths"], rtol=0, atol=0)
    close(a, g["rnd_angles"], rtol=0, atol=0)


def test_embeddings(golden):
    g = golden("g4_embeddings")
    for F in (8, 128):
        close(O.sinusoids_embedding(T(g["x"]), F), g[f"sin_F{F}"], rtol=0, atol=0)
    close(O.time_embedding(T(g["t"]), 256), g["time_256"], rtol=0, atol=0)


def _net_inputs(g):
    na = T(g["num_atoms"])
    n2g = torch.repeat_interleave(torch.arange(len(na)), na)
    return T(g["t_emb"]), T(g["atom_types"]), T(g["frac"]), T(g["lattices"]), na, n2g


def test_cspnet_tiny(golden):
    g = golden("g5a_cspnet_tiny")
    P = params(g)
    t_emb, at, fr, lat, na, n2g = _net_inputs(g)
    e = O.fc_edges(na)
    assert e.tolist() == g["edges"].tolist()
    close((fr[e[1]] - fr[e[0]]) % 1.0, g["frac_diff"], rtol=0, atol=0)
    taps = {}
    pl, px, pt = O.cspnet_forward(P, TINY, t_emb, at, fr, lat, na, n2g, taps=taps)
    for l in range(2):
        close(taps[f"h_{l}"], g[f"h_{l}"], rtol=1e-5, atol=1e-5)
    close(pl, g["pred_l"], rtol=1e-5, atol=1e-5)
    close(px, g["pred_x"], rtol=1e-5, atol=1e-5)
    close(pt, g["pred_t"], rtol=1e-5, atol=1e-5)


def test_cspnet_north_star_hparams(golden):
    """H=512, L=6, F=128: weights re-created from seed 0 (checksummed), outputs from the reference."""
    g = golden("g5b_cspnet_ns")
    hp = O.CSPNetHParams()
    P = O.init_params(hp, seed=0)
    assert sum(v.numel() for v in P.values()) == int(g["n_params"]) == 12346468
    for name, s, a in zip(g["param_names"].tolist(), g["param_sum"], g["param_abs_sum"]):
        assert abs(float(P[name].double().sum()) - s) <= 1e-9 * max(1.0, abs(a)), name
    close(P["decoder.csp_layer_3.edge_mlp.0.weight"][:4, :16], g["w_sample"], rtol=0, atol=0)
    t_emb, at, fr, lat, na, n2g = _net_inputs(g)
    pl, px, pt = O.cspnet_forward(P, hp, t_emb, at, fr, lat, na, n2g)
    close(pl, g["pred_l"], rtol=2e-5, atol=2e-5)
    close(px, g["pred_x"], rtol=2e-5, atol=2e-5)
    close(pt, g["pred_t"], rtol=2e-5, atol=2e-5)


def test_knn_branch(golden):
    g = golden("g5c_knn")
    P = params(g)
    t_emb, at, fr, lat, na, n2g = _net_inputs(g)
    cart = torch.einsum("bi,bij->bj", fr, lat[n2g])
    ei, jim, nb = O.radius_graph_pbc(cart, lat, na, 20)
    assert ei.tolist() == g["rg_edge_index"].tolist()
    assert jim.tolist() == g["rg_to_jimages"].tolist()
    assert nb.tolist() == g["rg_num_bonds"].tolist()
    edges, fd = O.knn_edges(fr, lat, na, 20)
    assert edges.tolist() == g["edges"].tolist()
    close(fd, g["edge_vec"], rtol=0, atol=0)
    hp = O.CSPNetHParams(hidden_dim=64, num_layers=2, num_freqs=8, edge_style="knn")
    pl, px, pt = O.cspnet_forward(P, hp, t_emb, at, fr, lat, na, n2g, edges=edges, frac_diff=fd)
    close(pl, g["pred_l"], rtol=1e-5, atol=1e-5)
    close(px, g["pred_x"], rtol=1e-5, atol=1e-5)
    close(pt, g["pred_t"], rtol=1e-5, atol=1e-5)


def _sched_from(g, Tn, key="P__sigma_scheduler.sigmas_norm"):
    return O.Schedules.make(Tn, sigmas_norm=T(np.asarray(g[key])))


def test_sample_chain(golden):
    g = golden("g6_sample")
    Tn = int(g["T"])
    P = params(g)
    sch = _sched_from(g, Tn)
    close(sch.beta["alphas_cumprod"], g["P__beta_scheduler.alphas_cumprod"], rtol=0, atol=0)
    na = T(g["num_atoms"])
    noise = dict(x_T=T(g["x_T"]), l_T=T(g["l_T"]), t_T=T(g["t_T"]), corr_x={}, pred_l={}, pred_t={}, pred_x={})
    for t in range(Tn, 1, -1):
        for k in ("corr_x", "pred_l", "pred_t", "pred_x"):
            noise[k][t] = T(g[f"n_{k}_{t}"])
    final, traj = O.sample(P, TINY, sch, na, noise, step_lr=float(g["step_lr"]))
    for t in range(Tn, -1, -1):
        # free-running 20-step chain: round-off compounds, so the bound is looser than one step
        for k in ("atom_types", "frac_coords", "lattices"):
            a, b = traj[t][k].numpy(), g[f"traj_{t}_{k}"]
            if k == "frac_coords":
                d = np.abs(a - b); d = np.minimum(d, 1 - d)
                assert d.max() < 2e-4, (t, k, d.max())
            else:
                np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4, err_msg=f"{t} {k}")
        if t > 1:
            for k in ("log_prob_l", "log_prob_t", "log_prob_x"):
                np.testing.assert_allclose(traj[t][k].numpy(), g[f"traj_{t}_{k}"], rtol=2e-3, atol=2e-3,
                                           err_msg=f"{t} {k}")


def test_add_noise_loss_kl(golden):
    g = golden("g7_noise_loss")
    P, Q = params(g), params(g, "Q__")
    sch = _sched_from(g, 1000, "sigmas_norm")
    batch = dict(num_atoms=T(g["num_atoms"]), lengths=T(g["lengths"]), angles=T(g["angles"]),
                 frac_coords=T(g["frac_coords"]), atom_types=T(g["atom_types"]))
    B = len(batch["num_atoms"])
    for ti in (0, 500, 999):
        noise = dict(rand_l=T(g[f"t{ti}_rand_l"]), rand_x=T(g[f"t{ti}_rand_x"]), rand_t=T(g[f"t{ti}_rand_t"]))
        noised = O.add_noise(TINY, sch, batch, ti, noise)
        (t_emb, atp, ifr, ilat, na, n2g), (rl, tar_x, rt), _ = noised
        close(t_emb, g[f"t{ti}_t_emb"], rtol=0, atol=0)
        close(atp, g[f"t{ti}_atom_type_probs"], rtol=0, atol=0)
        close(ifr, g[f"t{ti}_input_frac"], rtol=0, atol=0)
        close(ilat, g[f"t{ti}_input_lattice"], rtol=0, atol=0)
        close(tar_x, g[f"t{ti}_tar_x"], rtol=1e-6, atol=1e-6)
        loss, pred = O.calc_sample_loss(P, TINY, O.Costs(), noised)
        _, ppred = O.calc_sample_loss(Q, TINY, O.Costs(), noised)
        kl = O.calc_kl_reg(pred, ppred, n2g, B)
        close(loss, g[f"t{ti}_loss"], rtol=1e-5, atol=1e-5)
        close(kl, g[f"t{ti}_kl"], rtol=1e-5, atol=1e-6)
        close(pred[1], g[f"t{ti}_pred_x"], rtol=1e-5, atol=1e-5)


def test_ft_step(golden):
    g = golden("g8_ft_step")
    agent, prior = params(g), params(g, "Q__")
    sch = _sched_from(g, 1000, "sigmas_norm")
    batch = dict(num_atoms=T(g["num_atoms"]), lengths=T(g["lengths"]), angles=T(g["angles"]),
                 frac_coords=T(g["frac_coords"]), atom_types=T(g["atom_types"]))
    rewards = T(g["reward"])
    rec = {}
    noise_fn = lambda ep, t: dict(rand_l=T(g[f"s{t}_rand_l"]), rand_x=T(g[f"s{t}_rand_x"]), rand_t=T(g[f"s{t}_rand_t"]))
    # first optimizer step only: compare grads and params after it
    a1 = {k: v.clone() for k, v in agent.items()}
    O.ft_step(a1, prior, TINY, sch, O.Costs(), batch, rewards, noise_fn, lr=float(g["lr"]), timesteps=3,
              accum_steps=int(g["accum"]), sigma=float(g["sigma"]), record=rec)
    for t in range(3):
        close(rec["sample_loss"][t], g[f"s{t}_sample_loss"], rtol=1e-5, atol=1e-5)
        close(rec["kl"][t], g[f"s{t}_kl"], rtol=1e-4, atol=1e-7)
        close(rec["loss"][t], g[f"s{t}_loss"], rtol=1e-5, atol=1e-6)
    for k in agent:
        gr = g["G0__" + k]
        scale = max(1e-6, float(np.abs(gr).max()))
        np.testing.assert_allclose(rec["grads"][0][k].numpy(), gr, rtol=1e-3, atol=2e-5 * scale, err_msg=k)
        # Adam's first step moves every weight by ~lr regardless of |g|: compare the update, not the value
        np.testing.assert_allclose(a1[k].numpy(), g["A0__" + k], rtol=0, atol=2.5e-5, err_msg=k)
    # two optimizer steps (6 micro-steps)
    a2 = {k: v.clone() for k, v in agent.items()}
    O.ft_step(a2, prior, TINY, sch, O.Costs(), batch, rewards, noise_fn, lr=float(g["lr"]), timesteps=6,
              accum_steps=int(g["accum"]), sigma=float(g["sigma"]))
    for k in agent:
        np.testing.assert_allclose(a2[k].numpy(), g["A1__" + k], rtol=0, atol=5e-5, err_msg=k)


def test_host_glue(golden):
    g = golden("g9_host_glue")
    np.testing.assert_array_equal(np.array(O.ATOM_DIST_MP20), g["atom_dist_mp20"])
    assert O.DEFAULT_STEP_LR_GEN_MP20 == float(g["step_lr_gen_mp20"])
    np.random.seed(0)
    assert O.draw_num_atoms(8).tolist() == g["num_atoms_seed0_8"].tolist() == [10, 14, 12, 10, 8, 12, 9, 18]
    np.random.seed(0)
    assert O.draw_num_atoms(64).tolist() == g["num_atoms_seed0_64"].tolist()


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        r = O.philox4x32(np.array([ctr], dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(x) for x in r[0]) == out
    z = O.philox_normal(1234, 7, 3, 200001)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # offset-invariance: a shard sees the same numbers as the whole
    np.testing.assert_array_equal(O.philox_normal(1234, 7, 3, 100, elem_offset=37), z[37:137])
    u = O.philox_uniform(1234, 7, 0, 10001)
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def _csp_noise(g, tag, Tn):
    noise = dict(x_T=T(g[f"{tag}_x_T"]), l_T=T(g[f"{tag}_l_T"]), t_T=T(g[f"{tag}_t_T"]), corr_x={}, pred_l={}, pred_t={}, pred_x={})
    for t in range(Tn, 1, -1):
        for k in ("corr_x", "pred_l", "pred_t", "pred_x"):
            noise[k][t] = T(g[f"{tag}_n_{k}_{t}"])
    return noise


def test_sample_chain_csp_mode(golden):
    """keep_coords / keep_lattice (diffusion.py:283-287, 308-312, 330, 348-349) against the reference-generated chain."""
    g = golden("g10_csp_mode")
    Tn = int(g["T"])
    P = params(g)
    sch = _sched_from(g, Tn)
    for tag in ("kc", "kl"):
        na = T(g[f"{tag}_num_atoms"])
        noise = _csp_noise(g, tag, Tn)
        if tag == "kc":
            noise["x_T"] = T(g["kc_frac_coords"])
        else:
            noise["l_T"] = O.lattice_params_to_matrix(T(g["kl_lengths"]), T(g["kl_angles"]))
        final, traj = O.sample(P, TINY, sch, na, noise, step_lr=5e-6, keep_coords=tag == "kc", keep_lattice=tag == "kl")
        for t in range(Tn, -1, -1):
            for k in ("atom_types", "frac_coords", "lattices"):
                a, b = traj[t][k].numpy(), g[f"{tag}_traj_{t}_{k}"]
                if k == "frac_coords":
                    d = np.abs(a - b); d = np.minimum(d, 1 - d)
                    assert d.max() < 2e-4, (tag, t, k, d.max())
                else:
                    np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4, err_msg=f"{tag} {t} {k}")
            if t > 1:
                for k in ("log_prob_l", "log_prob_t", "log_prob_x"):
                    np.testing.assert_allclose(traj[t][k].numpy(), g[f"{tag}_traj_{t}_{k}"], rtol=2e-3, atol=2e-3, err_msg=f"{tag} {t} {k}")
        # the kept part really is the given one at the end of the chain
        if tag == "kc":
            d = np.abs(final["frac_coords"].numpy() - g["kc_frac_coords"] % 1.0)
            assert np.minimum(d, 1 - d).max() == 0
        else:
            assert np.array_equal(final["lattices"].numpy(), noise["l_T"].numpy())


def test_add_noise_with_sampled_times(golden):
    """add_noise(batch) without a time index (diffusion.py:83-84): per-crystal times from numpy's global generator."""
    g = golden("g11_noise_sampled_times")
    P = params(g)
    sch = _sched_from(g, 1000, "sigmas_norm")
    batch = dict(num_atoms=T(g["num_atoms"]), lengths=T(g["lengths"]), angles=T(g["angles"]), frac_coords=T(g["frac_coords"]),
                 atom_types=T(g["atom_types"]))
    noise = dict(rand_l=T(g["rand_l"]), rand_x=T(g["rand_x"]), rand_t=T(g["rand_t"]))
    np.random.seed(int(g["np_seed"]))
    noised = O.add_noise(TINY, sch, batch, None, noise)
    (t_emb, atp, ifr, ilat, na, n2g), (rl, tar_x, rt), _ = noised
    close(t_emb, g["t_emb"], rtol=0, atol=0)
    close(atp, g["atom_type_probs"], rtol=0, atol=0)
    close(ifr, g["input_frac"], rtol=0, atol=0)
    close(ilat, g["input_lattice"], rtol=0, atol=0)
    close(tar_x, g["tar_x"], rtol=1e-6, atol=1e-6)
    loss, pred = O.calc_sample_loss(P, TINY, O.Costs(), noised)
    close(loss, g["loss"], rtol=2e-5, atol=2e-5)
    # the explicit-times form the HIP tests use gives the same result
    n2 = O.add_noise(TINY, sch, batch, None, noise, times=g["times"])
    close(n2[0][2], g["input_frac"], rtol=0, atol=0)
