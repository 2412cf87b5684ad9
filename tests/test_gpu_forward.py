"""GPU parity of the HIP score network (through the C ABI) against the reference-generated
golden vectors and the CPU oracle.  fp32 tolerances are stated per assertion: the HIP path
re-associates the edge-MLP contraction (node-level projections + Fourier block), so results
agree to fp32 round-off, not bitwise."""
import numpy as np
import pytest
import torch

from oracle import diffcsp_oracle as O
from tests.gpu_util import load_decoder, params_from_golden

pytestmark = pytest.mark.gpu


# arithmetic paths: (gemm mode, edge mode).  Default = bf16-split GEMMs everywhere; the other two keep
# the f32-input MFMA (exact fp32 fma chains) with either edge-stage formulation.
# The default additionally evaluates the Fourier block once per unordered node pair (the reversed edge sees -sin / +cos of the
# same arguments); "per-edge" switches that off.
PATHS = [("split", "gemm"), ("split", "gemm", "per-edge"), ("f32", "gemm"), ("f32", "fused_f32")]


@pytest.fixture(params=PATHS, ids=lambda p: "-".join(p))
def path(request):
    from matinvent_amd.cspnet import set_edge_pairs, set_gemm_mode
    set_gemm_mode(request.param[0])
    set_edge_pairs(len(request.param) < 3)
    yield request.param
    set_gemm_mode("split")
    set_edge_pairs(True)


def _net(H, L, F, P=None, path=("split", "gemm")):
    from matinvent_amd.cspnet import CSPNet
    net = CSPNet(hidden_dim=H, num_layers=L, num_freqs=F, latent_dim=256, ln=True, smooth=True, pred_type=True, device="cuda")
    net.set_edge_mode(path[1])
    if P is not None:
        load_decoder(net, P)
    return net


def _close(a, b, tol, what):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:.1e} * {scale:.3g}"


def test_philox_matches_contract():
    import ctypes as C
    from matinvent_amd import _lib
    lib = _lib.load()
    for (step, draw, off, n, uni) in [(7, 3, 0, 4099, 0), (1001, 0, 37, 1000, 1), (5, 5, 4 * 123457 + 2, 777, 0)]:
        out = torch.empty(n, device="cuda")
        _lib.check(lib.mi_philox_fill(1234, step, draw, off, n, uni, C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize()
        ref = O.philox_uniform(1234, step, draw, n, off) if uni else O.philox_normal(1234, step, draw, n, off)
        # integer stream is exact; the Box-Muller transcendentals differ by libm round-off only
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=0 if uni else 5e-6)


def test_time_embedding_golden(golden):
    from matinvent_amd.diffcsp import SinusoidalTimeEmbeddings
    g = golden("g4_embeddings")
    emb = SinusoidalTimeEmbeddings(256).to("cuda")
    # own table (this host's torch.exp): 1-ulp table differences times t <= 1000 -> <= 6.2e-5
    _close(emb(torch.from_numpy(g["t"])), g["time_256"], 6.2e-5, "time embedding, local table")
    # the generating machine's table: device sin/cos agree to libm round-off
    emb.freqs.copy_(torch.from_numpy(g["time_freqs_256"]))
    _close(emb(torch.from_numpy(g["t"])), g["time_256"], 2e-6, "time embedding, pinned table")


@pytest.mark.parametrize("F", [8, 128])
def test_fourier_operand_golden(golden, F):
    """SinusoidsEmbedding (cspnet.py:12-24) as the edge stage consumes it: the pair-mode Fourier operand kernel, its two fp16 planes summed
    back to fp32, against the reference's embedding of the same differences (g4: `sin_F8`, `sin_F128`).  The product path never holds these
    values in fp32, so every other test sees them only through a forward.  Tolerance: 2^-21 of the unit range for the plane format (22
    significant bits at scale 2^6) + the device sin / cos against the generating host's libm (1.2e-7, common.h) + the argument's own
    rounding, which the reference shares only up to the order of its multiplications: 2 pi k d carries k <= 127 ulps of 2 pi d."""
    import ctypes as C
    from matinvent_amd import _lib
    g = golden("g4_embeddings")
    x = torch.from_numpy(g["x"]).float()
    n = x.shape[0]
    frac = torch.cat([torch.zeros(1, 3), x]).cuda().contiguous()            # atom 0 at the origin: d(0, j) = x_j % 1
    pi = torch.zeros(n, dtype=torch.int32, device="cuda")
    pj = torch.arange(1, n + 1, dtype=torch.int32, device="cuda")
    out = torch.empty(n, 6 * F, device="cuda")
    _lib.check(_lib.load().mi_debug_fourier_pairs(C.c_void_p(frac.data_ptr()), C.c_void_p(pi.data_ptr()), C.c_void_p(pj.data_ptr()), n, F,
                                                  C.c_void_p(out.data_ptr()), None), "mi_debug_fourier_pairs")
    ref = torch.from_numpy(g[f"sin_F{F}"])
    err = (out.cpu() - ref).abs()
    k = torch.arange(F).repeat(6).float()                                   # frequency of every column
    bound = 2.0 ** -21 + 1.2e-7 + (k + 1) * 2 * torch.pi * 2.0 ** -24 * 1.5
    assert bool((err <= bound[None, :]).all()), f"F={F}: max err {float(err.max()):.3e} at column {int(err.max(0).values.argmax())}"


def test_forward_tiny_golden(golden, path):
    g = golden("g5a_cspnet_tiny")
    net = _net(64, 2, 8, params_from_golden(g), path)
    T = lambda k: torch.from_numpy(g[k]).cuda()
    b = net.make_batch(g["num_atoms"])
    assert b.num_edges == g["edges"].shape[1]
    pl, px, pt = net(T("t_emb"), T("atom_types"), T("frac"), T("lattices"), None, batch=b)
    for l in range(2):
        _close(net.tap(b, l + 1), g[f"h_{l}"], 2e-5, f"h after layer {l}")
    _close(pl, g["pred_l"], 2e-5, "pred_l")
    _close(px, g["pred_x"], 2e-5, "pred_x")
    _close(pt, g["pred_t"], 2e-5, "pred_t")


def test_forward_north_star_hparams_golden(golden, path):
    """H=512, L=6, F=128: weights rebuilt from seed 0 (same construction order as the
    reference, checksummed), outputs produced by the reference code."""
    g = golden("g5b_cspnet_ns")
    torch.manual_seed(0)
    net = _net(512, 6, 128, path=path)
    sd = net.state_dict()
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"])
    for name, s, a in zip(g["param_names"].tolist(), g["param_sum"], g["param_abs_sum"]):
        assert abs(float(sd[name[len("decoder."):]].double().sum()) - s) <= 1e-9 * max(1.0, abs(a)), name
    T = lambda k: torch.from_numpy(g[k]).cuda()
    pl, px, pt = net(T("t_emb"), T("atom_types"), T("frac"), T("lattices"), g["num_atoms"])
    _close(pl, g["pred_l"], 5e-5, "pred_l")
    _close(px, g["pred_x"], 5e-5, "pred_x")
    _close(pt, g["pred_t"], 5e-5, "pred_t")


@pytest.mark.parametrize("H,L,F,num_atoms", [
    (64, 2, 8, [1]),
    (64, 2, 10, [1, 2, 3, 20, 7, 1, 13]),         # F=10: Fourier pairs padded 30 -> 32
    (128, 3, 10, [5, 0, 9, 20, 20, 4]),           # an empty crystal
    (256, 2, 16, [20] * 9 + [3]),
    (64, 1, 8, [40, 33, 2]),                      # node runs spanning three 32-edge tiles
    (192, 2, 10, [5, 0, 9, 20, 20, 4]),           # widths that are multiples of 64 but not powers of two: plane GEMMs throughout
    (384, 2, 16, [20] * 9 + [3]),
])
def test_forward_vs_oracle_ragged(H, L, F, num_atoms, path):
    if H not in (64, 128, 256, 512) and path[1] == "fused_f32":
        pytest.skip("the register-chained f32 edge kernel exists for hidden_dim 64/128/256/512 only (mi_net_set_edge_mode refuses the others)")
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=3)
    # non-trivial LayerNorm affine so that its parameters are exercised
    g = torch.Generator().manual_seed(11)
    for k in P:
        if "layer_norm" in k:
            P[k] = P[k] + 0.1 * torch.randn(P[k].shape, generator=g)
    net = _net(H, L, F, P, path)
    na = torch.tensor(num_atoms)
    B, N = len(num_atoms), int(na.sum())
    n2g = torch.repeat_interleave(torch.arange(B), na)
    t_emb = O.time_embedding(torch.full((B,), 321), 256)
    at = torch.randn(N, 100, generator=g)
    fr = torch.rand(N, 3, generator=g) * 3 - 1  # also outside [0,1): the sampler feeds unwrapped x_{t-1/2}
    lat = torch.randn(B, 3, 3, generator=g) * 2
    ol, ox, ot = O.cspnet_forward(P, hp, t_emb, at, fr, lat, na, n2g)
    pl, px, pt = net(t_emb.cuda(), at.cuda(), fr.cuda(), lat.cuda(), na)
    _close(pl, ol, 3e-5, "pred_l")
    _close(px, ox, 3e-5, "pred_x")
    _close(pt, ot, 3e-5, "pred_t")


def test_forward_full_size_properties(path):
    """BASELINE config-2 shape (B=256, n=20, H=512, L=6, F=128): size-independent properties.
    (1) bit-reproducible run to run (fixed reduction order, no float atomics);
    (2) a crystal's outputs do not depend on which other crystals share the batch
        (first 32 crystals evaluated alone agree with the full batch to fp32 round-off);
    (3) a lattice-vector translation of all atoms of a crystal leaves the scores unchanged
        up to round-off (only (x_j - x_i) % 1 enters the network)."""
    torch.manual_seed(0)
    net = _net(512, 6, 128, path=path)
    B, n = 256, 20
    g = torch.Generator().manual_seed(5)
    na = [n] * B
    N = B * n
    t_emb = O.time_embedding(torch.full((B,), 500), 256).cuda()
    at = torch.randn(N, 100, generator=g).cuda()
    fr = torch.rand(N, 3, generator=g).cuda()
    lat = torch.randn(B, 3, 3, generator=g).cuda()
    bt = net.make_batch(na)
    o1 = [x.clone() for x in net(t_emb, at, fr, lat, None, batch=bt)]
    o2 = net(t_emb, at, fr, lat, None, batch=bt)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b), "forward is not bit-reproducible"
    for a in o1:
        assert torch.isfinite(a).all()
    k = 32
    s = net(t_emb[:k], at[:k * n], fr[:k * n], lat[:k], [n] * k)
    _close(s[0], o1[0][:k], 1e-5, "pred_l shard")
    _close(s[1], o1[1][:k * n], 1e-5, "pred_x shard")
    _close(s[2], o1[2][:k * n], 1e-5, "pred_t shard")
    shift = torch.rand(B, 1, 3, generator=g).cuda().expand(B, n, 3).reshape(N, 3)
    o3 = net(t_emb, at, (fr + shift) % 1.0, lat, None, batch=bt)
    for a, b, w in zip(o1, o3, ("pred_l", "pred_x", "pred_t")):
        _close(b, a, 2e-4, w + " translation")


def test_full_size_paths_agree():
    """BASELINE config-2 shape: the default path (pre-split bf16 planes, 256x128 double-buffered GEMM with the row-major and the
    fused segmented-sum epilogues) against the independent f32-input MFMA path on the same inputs -- the full-size check of the
    kernels the small oracle cases cannot reach (they only engage above ~33k edges)."""
    from matinvent_amd.cspnet import set_gemm_mode
    torch.manual_seed(0)
    B, n = 256, 20
    g = torch.Generator().manual_seed(11)
    N = B * n
    t_emb = O.time_embedding(torch.full((B,), 321), 256).cuda()
    at = torch.randn(N, 100, generator=g).cuda()
    fr = torch.rand(N, 3, generator=g).cuda()
    lat = (4 * torch.eye(3) + torch.randn(B, 3, 3, generator=g)).cuda()
    outs = {}
    net = _net(512, 6, 128, path=("split", "gemm"))
    bt = net.make_batch([n] * B)
    for mode in ("split", "f32"):
        set_gemm_mode(mode)
        try:
            outs[mode] = [x.clone() for x in net(t_emb, at, fr, lat, None, batch=bt)]
        finally:
            set_gemm_mode("split")
    for a, b, w in zip(outs["split"], outs["f32"], ("pred_l", "pred_x", "pred_t")):
        _close(a, b, 2e-5, w + " split vs f32")


def test_full_size_forward_vs_oracle():
    """BASELINE config-2 shape (B=256, n=20, H=512, L=6, F=128, E=102 400): the default HIP path against the CPU oracle itself
    (one evaluation; the oracle materialises the reference's [E, 1801] edge input, ~0.7 GB).  This is the direct parity check of
    the kernels that only engage at full size: pair-mode Fourier GEMM, 256-row double-buffered GEMM, fused segmented sum."""
    B, n, H, L, F = 256, 20, 512, 6, 128
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=4)
    net = _net(H, L, F, P)
    g = torch.Generator().manual_seed(17)
    N = B * n
    na = torch.full((B,), n, dtype=torch.long)
    n2g = torch.repeat_interleave(torch.arange(B), na)
    t_emb = O.time_embedding(torch.full((B,), 640), 256)
    at = torch.randn(N, 100, generator=g)
    fr = torch.rand(N, 3, generator=g)
    lat = 4 * torch.eye(3) + torch.randn(B, 3, 3, generator=g)
    with torch.no_grad():
        ref = O.cspnet_forward(P, hp, t_emb, at, fr, lat, na, n2g)
    out = net(t_emb.cuda(), at.cuda(), fr.cuda(), lat.cuda(), None, batch=net.make_batch([n] * B))
    for a, b, w in zip(out, ref, ("pred_l", "pred_x", "pred_t")):
        _close(a, b, 2e-5, w + " full size vs oracle")


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_forward_random_ragged_batches_vs_oracle(seed):
    """Randomly drawn ragged batches (1..30 atoms, empty crystals allowed, skewed cells) on the default path: pair tables,
    segment runs crossing tile boundaries, crystals larger than a 32-row block."""
    rng = np.random.RandomState(seed)
    Bc = int(rng.randint(1, 13))
    na = [int(x) for x in rng.randint(0, 31, size=Bc)]
    if sum(na) == 0:
        na[0] = 3
    H, L, F = (64, 2, 8) if seed % 2 else (128, 1, 10)
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=seed)
    net = _net(H, L, F, P)
    g = torch.Generator().manual_seed(seed)
    N = sum(na)
    nat = torch.tensor(na)
    n2g = torch.repeat_interleave(torch.arange(Bc), nat)
    t_emb = O.time_embedding(torch.randint(1, 1000, (Bc,), generator=g), 256)
    at, fr = torch.randn(N, 100, generator=g), torch.rand(N, 3, generator=g)
    lat = 3 * torch.eye(3) + torch.randn(Bc, 3, 3, generator=g)
    ref = O.cspnet_forward(P, hp, t_emb, at, fr, lat, nat, n2g)
    out = net(t_emb.cuda(), at.cuda(), fr.cuda(), lat.cuda(), None, batch=net.make_batch(na))
    for a, b, w in zip(out, ref, ("pred_l", "pred_x", "pred_t")):
        _close(a, b, 2e-5, f"{w} seed {seed} atoms {na}")


def test_inference_heads_kernel_vs_golden_and_vs_the_two_gemm_heads(golden):
    """The inference forward forms the coordinate and type heads in one launch (plain fp32 FMA chains over the final LayerNorm's rows);
    the training forward keeps the two fp32-operand GEMM launches.  Both against the reference-generated outputs at the benchmark
    hyper-parameters, and against each other."""
    from matinvent_amd import _lib
    g = golden("g5b_cspnet_ns")
    torch.manual_seed(0)
    net = _net(512, 6, 128)
    T = lambda k: torch.from_numpy(g[k]).cuda()
    outs = []
    try:
        for knob in (0, 2):
            _lib.check(_lib.load().mi_debug_set_node_priority(knob))
            with torch.no_grad():
                outs.append(net(T("t_emb"), T("atom_types"), T("frac"), T("lattices"), g["num_atoms"]))
    finally:
        _lib.check(_lib.load().mi_debug_set_node_priority(0))
    for pl, px, pt in outs:
        _close(pl, g["pred_l"], 5e-5, "pred_l")
        _close(px, g["pred_x"], 5e-5, "pred_x")
        _close(pt, g["pred_t"], 5e-5, "pred_t")
    assert torch.equal(outs[0][0], outs[1][0])   # (the lattice head is the same kernel)
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("H,L,F,num_atoms,style", [
    (128, 3, 10, [5, 0, 9, 20, 20, 4, 1], "fc"),          # 59 nodes: a partial last 32-row block
    (256, 2, 16, [20] * 9 + [3], "fc"),
    (512, 2, 16, [20, 7, 1, 33, 12, 20, 20, 2], "fc"),   # both wave forms of the 512-wide kernel
    (512, 2, 16, [20] * 40, "fc"),                        # 800 nodes: the seven-launch form runs its node products on the plane GEMMs
    (512, 1, 16, [1] * 150 + [2] * 20 + [40, 3], "fc"),   # one-edge nodes: 128 local nodes in a 128-row tile (four S blocks), and a 40-atom crystal
    (256, 2, 16, [12, 20, 8, 16], "knn"),
    (512, 2, 16, [12, 20, 8, 16, 20, 5], "knn"),
])
def test_node_chain_launch_vs_the_seven_launch_form(H, L, F, num_atoms, style):
    """Inference forwards run everything between two edge stages (segmented mean, node MLP + residual, LayerNorm, the projections
    LayerNorm(h) feeds: cspnet.py:79-91,61) as one launch per layer boundary (csrc/node_chain.hip).  Same plane format, same scales, same
    k order as the plane GEMMs it replaces: outputs and every layer's node features agree to a few ulp of the accumulations -- and both
    forms agree with the oracle at the forward tests' tolerance."""
    from matinvent_amd import _lib
    lib = _lib.load()
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=21)
    g = torch.Generator().manual_seed(23)
    for k in P:
        if "layer_norm" in k:
            P[k] = P[k] + 0.1 * torch.randn(P[k].shape, generator=g)
    from matinvent_amd.cspnet import CSPNet
    net = CSPNet(hidden_dim=H, num_layers=L, num_freqs=F, latent_dim=256, ln=True, smooth=True, pred_type=True, device="cuda", edge_style=style)
    load_decoder(net, P)
    na = torch.tensor(num_atoms)
    B, N = len(num_atoms), int(na.sum())
    t_emb = O.time_embedding(torch.full((B,), 77), 256).cuda()
    at = torch.randn(N, 100, generator=g).cuda()
    fr = torch.rand(N, 3, generator=g).cuda()
    lat = (5 * torch.eye(3) + 0.5 * torch.randn(B, 3, 3, generator=g)).cuda()
    bt = net.make_batch(num_atoms)
    res = {}
    try:
        # knob: node chain as seven launches (0) / one launch (1) / one launch with the deeper weight ring (2); 3 = one launch AND the second
        # edge GEMM on the 128-row x H-column register tiles with the segmented sum on the matrix pipe (edge_stage.hip; hidden_dim 512 only)
        # 4 = form 3 AND the pair-mode first edge GEMM on the same form (weights in fragment order straight from L2); 5 = its 128 x 256-tile
        # variant (one workgroup per CU, 512 registers per lane; an ablation build only, otherwise form 4 again); 6 = its 2 x 2-wave variant
        # (a wave owns 64 pairs x 64 columns); 7 = form 3 with BOTH edge products and the edge -> node sums in one launch (edge_fused.hip: a
        # workgroup owns 64 pairs, M1 stays in LDS; hidden_dim 512, fc; exists in ablation builds only -- the default library runs form 3 again);
        # 8 = form 3 with the pair-mode first edge GEMM on ONE accumulator set (128 x 256 tiles, the sine half of K a second time against -2 Wsin:
        # edge_gemm1e_kernel; widths that are multiples of 256, otherwise form 4's kernel)
        # 9 = form 3 with the pair-mode epilogue's 64-bit addressing (taken by itself only beyond 4 GB of operands)
        # 10 / 11 = form 3 with the chain's products COLUMN-SPLIT over workgroups (node_cols_kernel: 32 rows x 128 columns of one product per
        # workgroup) as one launch per stage / as one launch per layer boundary with agent-scope flag hand-overs; forms 1 .. 9 keep the row-block chain
        for knob in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
            lib.mi_debug_set_node_cols(knob - 9 if knob >= 10 else 0)
            lib.mi_debug_set_pair_wide(1 if knob == 9 else 0)
            lib.mi_debug_set_node_fused(1 if knob >= 3 else knob)
            lib.mi_debug_set_edge2_fused(1 if knob >= 3 else 0)
            lib.mi_debug_set_edge1_fused(knob - 3 if 4 <= knob <= 6 else 4 if knob == 8 else 0)
            lib.mi_debug_set_edge_fused(1 if knob == 7 else 0)
            with torch.no_grad():
                outs = [x.clone() for x in net(t_emb, at, fr, lat, None, batch=bt)]
            res[knob] = outs + [net.tap(bt, l + 1).clone() for l in range(L)]
            assert all(torch.isfinite(x).all() for x in res[knob])
    finally:
        lib.mi_debug_set_node_fused(1)
        lib.mi_debug_set_node_cols(3)
        lib.mi_debug_set_edge2_fused(1)
        lib.mi_debug_set_edge1_fused(9)   # (the default: the register-tile form for launches beyond the plane GEMM's small-launch forms)
        lib.mi_debug_set_edge_fused(0)
        lib.mi_debug_set_pair_wide(0)
    assert _lib.saturation_events(reset=True) == 0
    for knob in (4, 5, 6, 9):
        for a, b, w in zip(res[knob], res[3], ["pred_l", "pred_x", "pred_t"]):   # same epilogue, same k and term order: the same M1, bit for bit
            assert torch.equal(a, b), f"{w}: the first edge GEMM's forms differ (form {knob})"
    for knob in (10, 11):   # the column-split chain: same arithmetic, same k and term order per output element -- every layer's features, bit for bit
        for a, b, w in zip(res[knob], res[3], ["pred_l", "pred_x", "pred_t"] + [f"h after layer {l}" for l in range(L)]):
            assert torch.equal(a, b), f"{w}: the column-split node chain (form {knob}) differs from the row-block launch"
    names = ["pred_l", "pred_x", "pred_t"] + [f"h after layer {l}" for l in range(L)]
    for knob in (1, 2, 3, 4, 5, 6, 7, 8):
        for a, b, w in zip(res[knob], res[0], names):
            # (form 3 sums M2 rounded to the plane format's 22 bits: 1e-6 instead of a few ulp)
            _close(a, b, 2e-6 if knob < 3 else 5e-6, f"{w}: form {knob} vs seven launches")
    if style == "fc":
        n2g = torch.repeat_interleave(torch.arange(B), na)
        with torch.no_grad():
            ref = O.cspnet_forward(P, hp, t_emb.cpu(), at.cpu(), fr.cpu(), lat.cpu(), na, n2g)
        for knob in (1, 4, 7, 8):
            for a, b, w in zip(res[knob][:3], ref, names):
                _close(a, b, 3e-5, f"{w} (form {knob}) vs oracle")
