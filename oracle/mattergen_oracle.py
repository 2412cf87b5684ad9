"""CPU oracle of the MatterGen-shaped path (TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke and
bench.py's cpu_baseline leg -- never by the product path).

PARITY UNPINNED.  The reference ships only a 571-line adapter for MatterGen (models/mattergen/{pl_module,loss,sample,
dataset}.py, models/suite/mattergen.py); the denoiser (GemNet-T), the corruptions, the D3PM loss and the predictor-corrector
sampler live in the un-vendored dependency `mattergen @ git 5bb2b397a36de85a8dc9583b7d1d6353989de72c` (env.yml:31), which is
absent from /root/reference and cannot be installed here.  What follows restates the PUBLISHED architecture and algorithms
[UPSTREAM-UNVERIFIED] (GemNet-T: Gasteiger et al. 2021, as used by CDVAE / MatterGen: Zeni et al. 2025) with every
definition written out, so that the HIP path has one precise statement to be checked against.  Results on this path are
"self-consistent, parity-unpinned vs upstream".  What IS pinned by the reference's own files:
  * the adapter surface and its arithmetic: the time grid `linspace(T_max, 1/1000, 1000)[timestep]` (pl_module.py:55-69), the
    per-sample loss `sum_field w_field * loss_field` with w = {atomic_numbers 1, cell 1, pos 0.1} (loss.py:22-26, 71-73), the
    anchor penalty (pl_module.py:83-102), the sampler returning the `mean` batch (sample.py:49-50), the dataset fields
    (dataset.py:42-65).

Shapes: B crystals, N atoms, E directed edges (symmetric list), fractional positions x [N,3], cell L [B,3,3] (rows = lattice
vectors, cart = frac @ L), atomic numbers a [N] in 1..100, MASK = 101 (D3PM absorbing state, class index = a - 1).
"""
import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

NUM_CLASSES = 101           # 100 elements + the D3PM mask state
MASK = 101                  # atomic number standing for the absorbing state
INV_SQRT_2 = 1.0 / math.sqrt(2.0)
ACT_SCALE = 1.0 / 0.6       # ScaledSiLU


@dataclass
class GemNetHParams:
    emb_atom: int = 512
    emb_edge: int = 512
    emb_trip: int = 64
    emb_rbf: int = 16
    emb_cbf: int = 16
    emb_bil: int = 64
    num_radial: int = 128
    num_spherical: int = 7
    num_blocks: int = 4
    num_before_skip: int = 1
    num_after_skip: int = 2
    num_concat: int = 1
    num_atom: int = 3
    cutoff: float = 7.0
    max_neighbors: int = 50
    max_images: int = 5      # periodic images per dimension on each side


TINY = dict(emb_atom=64, emb_edge=64, emb_trip=32, emb_rbf=8, emb_cbf=8, emb_bil=32, num_radial=16, num_spherical=4, num_blocks=2,
            num_before_skip=1, num_after_skip=1, num_concat=1, num_atom=1, cutoff=5.0, max_neighbors=8, max_images=3)


# --------------------------------------------------------------------------------------------------------------------------
# parameters: ONE ordered list of (name, rows, cols); the HIP library exposes the same list (mi_gemnet_param_info)
# --------------------------------------------------------------------------------------------------------------------------
def param_list(hp: GemNetHParams):
    A, Ed, Tr, Rb, Cb, Bl, R, S = hp.emb_atom, hp.emb_edge, hp.emb_trip, hp.emb_rbf, hp.emb_cbf, hp.emb_bil, hp.num_radial, hp.num_spherical
    out = [("atom_emb.weight", NUM_CLASSES, A), ("atom_latent_emb.weight", A, 2 * A), ("atom_latent_emb.bias", 1, A),
           ("edge_emb.weight", Ed, 2 * A + R), ("mlp_rbf3.weight", Rb, R), ("mlp_cbf3.weight", S * Cb, R), ("mlp_rbf_h.weight", Rb, R),
           ("mlp_rbf_out.weight", Rb, R)]

    def res(prefix, n, width):
        r = []
        for k in range(n):
            r += [(f"{prefix}.{k}.0.weight", width, width), (f"{prefix}.{k}.1.weight", width, width)]
        return r

    def out_block(i):
        # GemNet-T's OutputBlock with direct forces [UPSTREAM-UNVERIFIED]: an ENERGY path (radial weighting, edge -> atom sum, ScalingFactor,
        # Dense + num_atom residual layers, a one-column read-out: `E_t`, which the MatterGen denoiser computes and never reads) and a FORCE
        # path (Dense + num_atom residual layers on the edge embedding, radial weighting, ScalingFactor, a one-column read-out); the
        # lattice head (dense_S / rbf_S / out_S) is this restatement's form of the per-block lattice update
        p = f"out_blocks.{i}"
        return ([(f"{p}.dense_rbf.weight", Ed, Rb), (f"{p}.scale_sum.scale_factor", 1, 1), (f"{p}.seq_energy.dense.weight", A, Ed)]
                + res(f"{p}.seq_energy.res", hp.num_atom, A) + [(f"{p}.out_energy.weight", 1, A)]
                + [(f"{p}.dense_F.weight", Ed, Ed)] + res(f"{p}.res_F", hp.num_atom, Ed)
                + [(f"{p}.rbf_F.weight", Ed, Rb), (f"{p}.scale_rbf_F.scale_factor", 1, 1), (f"{p}.out_F.weight", 1, Ed),
                   (f"{p}.dense_S.weight", Ed, Ed), (f"{p}.rbf_S.weight", Ed, Rb), (f"{p}.out_S.weight", 1, Ed)])

    out += out_block(0)
    for i in range(hp.num_blocks):
        p = f"int_blocks.{i}"
        out += [(f"{p}.dense_ca.weight", Ed, Ed), (f"{p}.dense_ba.weight", Ed, Ed), (f"{p}.mlp_rbf.weight", Ed, Rb), (f"{p}.scale_rbf.scale_factor", 1, 1),
                (f"{p}.down_projection.weight", Tr, Ed), (f"{p}.bilinear.weight", Bl, Cb * Tr), (f"{p}.scale_cbf_sum.scale_factor", 1, 1),
                (f"{p}.up_projection_ca.weight", Ed, Bl), (f"{p}.up_projection_ac.weight", Ed, Bl)]
        out += res(f"{p}.before_skip", hp.num_before_skip, Ed) + res(f"{p}.after_skip", hp.num_after_skip, Ed)
        out += [(f"{p}.atom_update.rbf.weight", Ed, Rb), (f"{p}.atom_update.scale_sum.scale_factor", 1, 1), (f"{p}.atom_update.dense.weight", A, Ed)]
        out += res(f"{p}.atom_update.res", hp.num_atom, A)
        out += [(f"{p}.concat.weight", Ed, 2 * A + Ed)] + res(f"{p}.residual_m", hp.num_concat, Ed)
        out += out_block(i + 1)
    out += [("fc_atom.weight", NUM_CLASSES, A), ("fc_atom.bias", 1, NUM_CLASSES)]
    return out


def init_params(hp: GemNetHParams, seed: int = 0, head_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Variance-preserving normal init (std = 1/sqrt(fan_in)); biases zero; the per-edge output heads scaled by head_scale."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, rows, cols in param_list(hp):
        if name.endswith(".bias"):
            P[name] = torch.zeros(cols)
        elif name.endswith(".scale_factor"):   # GemNet's ScalingFactor: a fitted constant upstream (a buffer, never trained); identity here
            P[name] = torch.ones(1, 1)
        elif name == "atom_emb.weight":
            P[name] = torch.randn(rows, cols, generator=g)
        else:
            P[name] = torch.randn(rows, cols, generator=g) / math.sqrt(cols)
            if ".out_F." in name or ".out_S." in name or ".out_energy." in name or name == "fc_atom.weight":
                P[name] = P[name] * head_scale
    return P


def param_grads(loss, A: Dict[str, torch.Tensor], retain_graph: bool = False):
    """d loss / d A as a list in A's order; entries no output depends on (the energy path, the ScalingFactor constants) are zeros."""
    gs = torch.autograd.grad(loss, list(A.values()), allow_unused=True, retain_graph=retain_graph)
    return [torch.zeros_like(v) if g is None else g for g, v in zip(gs, A.values())]


def is_scale_factor(name: str) -> bool:
    """The non-trainable scalars of the parameter list (their gradient entries are zero by definition)."""
    return name.endswith(".scale_factor")


# --------------------------------------------------------------------------------------------------------------------------
# graph: periodic radius graph, max_neighbors nearest per target atom, symmetrised (the construction GemNet-T needs: every edge
# c->a has its reverse a->c in the list; OCP `radius_graph_pbc` + `select_symmetric_edges`, [UPSTREAM-UNVERIFIED])
# --------------------------------------------------------------------------------------------------------------------------
def build_graph(frac: torch.Tensor, cell: torch.Tensor, num_atoms: torch.Tensor, hp: GemNetHParams):
    """Returns dict(src, dst, img [E,3] int, rowptr [N+1], swap [E], D [E], V [E,3], edge_graph [E]); edges sorted by target atom
    `dst`, within a target by (source, image code).  Edge e = (src c -> dst a) with vector r_c + img @ L - r_a, D = its length,
    V = unit vector.  All arithmetic in fp32 with the operation order the device kernel uses.
      1. images per dimension: rep_k = min(max_images, ceil(cutoff / plane spacing_k));
      2. candidates of target a: all (c, img) with 1e-6 < d^2 <= cutoff^2;
      3. keep the max_neighbors smallest in the order (d^2, c, image code);
      4. symmetrise: a pair is represented by its "upper" member -- c < a, or c == a with image code < the code of (0,0,0) -- and is
         kept iff the upper member was selected by its target; both directions are then emitted."""
    f32 = torch.float32
    frac, cell = frac.to(f32), cell.to(f32)
    B = int(num_atoms.shape[0])
    off = [0] + torch.cumsum(num_atoms, 0).tolist()
    src_l, dst_l, img_l = [], [], []
    R = hp.max_images
    W = 2 * R + 1
    r2 = np.float32(hp.cutoff) * np.float32(hp.cutoff)
    for b in range(B):
        n0, n1 = off[b], off[b + 1]
        n = n1 - n0
        if n == 0:
            continue
        L = cell[b].numpy().astype(np.float32)
        f = frac[n0:n1].numpy().astype(np.float32)
        # cart = frac @ L as an fma chain (f0*L0 + f1*L1 + f2*L2, innermost product first)
        cart = np.zeros((n, 3), np.float32)
        for c in range(3):
            acc = (f[:, 0].astype(np.float64) * L[0, c]).astype(np.float32)
            acc = (f[:, 1].astype(np.float64) * L[1, c] + acc).astype(np.float32)
            acc = (f[:, 2].astype(np.float64) * L[2, c] + acc).astype(np.float32)
            cart[:, c] = acc
        c23, c31, c12 = np.cross(L[1], L[2]), np.cross(L[2], L[0]), np.cross(L[0], L[1])
        vol = abs(float(np.dot(L[0].astype(np.float64), c23.astype(np.float64))))
        reps = []
        for cr in (c23, c31, c12):
            nrm = float(np.sqrt((cr.astype(np.float64) ** 2).sum()))
            spacing = vol / max(nrm, 1e-30)
            reps.append(int(min(R, max(1, math.ceil(hp.cutoff / max(spacing, 1e-30) - 1e-9)))))
        ia, ib, ic = np.meshgrid(np.arange(-reps[0], reps[0] + 1), np.arange(-reps[1], reps[1] + 1), np.arange(-reps[2], reps[2] + 1),
                                 indexing="ij")
        imgs = np.stack([ia.ravel(), ib.ravel(), ic.ravel()], 1)                                  # [I,3]
        codes = (imgs[:, 0] + R) * W * W + (imgs[:, 1] + R) * W + (imgs[:, 2] + R)
        # offs = ia*L0 + ib*L1 + ic*L2 in fp32: (ia*L0 + ib*L1) + ic*L2
        o = (imgs[:, 0:1].astype(np.float32) * L[0][None] + imgs[:, 1:2].astype(np.float32) * L[1][None]).astype(np.float32)
        o = (o + imgs[:, 2:3].astype(np.float32) * L[2][None]).astype(np.float32)                 # [I,3]
        zero_code = R * W * W + R * W + R
        sel_upper = set()
        for a in range(n):
            p = (cart[None, :, :] + o[:, None, :]).astype(np.float32)                             # [I,n,3] position of (c, img)
            d = (p - cart[a][None, None, :]).astype(np.float32)
            d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]).astype(np.float32)
            ii, cc = np.nonzero((d2 <= r2) & (d2 > np.float32(1e-6)))
            keys = sorted(zip(d2[ii, cc].tolist(), cc.tolist(), codes[ii].tolist()))[: hp.max_neighbors]
            for _, c, code in keys:
                if c < a or (c == a and code < zero_code):
                    sel_upper.add((a, c, code))
        edges = []
        for (a, c, code) in sel_upper:
            im = (code // (W * W) - R, (code // W) % W - R, code % W - R)
            edges.append((a, c, code, im))                                                        # c -> a
            rcode = (-im[0] + R) * W * W + (-im[1] + R) * W + (-im[2] + R)
            edges.append((c, a, rcode, (-im[0], -im[1], -im[2])))                                # a -> c
        edges.sort(key=lambda t: (t[0], t[1], t[2]))
        for a, c, code, im in edges:
            dst_l.append(n0 + a)
            src_l.append(n0 + c)
            img_l.append(im)
    N = off[-1]
    E = len(src_l)
    src = torch.tensor(src_l, dtype=torch.long)
    dst = torch.tensor(dst_l, dtype=torch.long)
    img = torch.tensor(img_l, dtype=torch.long).reshape(E, 3)
    rowptr = torch.zeros(N + 1, dtype=torch.long)
    if E:
        rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=N), 0)
    key = {(int(s), int(d), tuple(i.tolist())): e for e, (s, d, i) in enumerate(zip(src, dst, img))}
    swap = torch.tensor([key[(int(d), int(s), tuple((-i).tolist()))] for s, d, i in zip(src, dst, img)], dtype=torch.long).reshape(E)
    n2g = torch.repeat_interleave(torch.arange(B), num_atoms)
    g = dict(src=src, dst=dst, img=img, rowptr=rowptr, swap=swap, edge_graph=n2g[dst] if E else torch.zeros(0, dtype=torch.long), node2graph=n2g)
    g.update(edge_geometry(frac, cell, g))
    return g


def edge_geometry(frac, cell, g):
    """D [E], V [E,3] of the edge list (differentiable in torch, though the fine-tune step never needs input gradients)."""
    n2g = g["node2graph"]
    cart = torch.einsum("ni,nij->nj", frac, cell[n2g])
    vec = cart[g["src"]] + torch.einsum("ei,eij->ej", g["img"].to(frac.dtype), cell[g["edge_graph"]]) - cart[g["dst"]]
    D = torch.sqrt((vec * vec).sum(-1))
    return dict(D=D, V=vec / D[:, None])


# --------------------------------------------------------------------------------------------------------------------------
# basis functions
# --------------------------------------------------------------------------------------------------------------------------
def radial_basis(D: torch.Tensor, hp: GemNetHParams) -> torch.Tensor:
    """Polynomial envelope (exponent 5) x Gaussian smearing on d = D / cutoff in [0, 1]: [E, num_radial]."""
    d = D / hp.cutoff
    p = 5
    a, b, c = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
    env = 1 + a * d ** p + b * d ** (p + 1) + c * d ** (p + 2)
    env = torch.where(d < 1, env, torch.zeros_like(d))
    offset = torch.linspace(0, 1, hp.num_radial)
    coeff = -0.5 / float(offset[1] - offset[0]) ** 2
    return env[:, None] * torch.exp(coeff * (d[:, None] - offset[None, :]) ** 2)


def legendre_sph(cos: torch.Tensor, S: int) -> torch.Tensor:
    """Real spherical harmonics Y_l^0(cos) = sqrt((2l+1)/(4 pi)) P_l(cos), l < S: [..., S]."""
    P = [torch.ones_like(cos), cos]
    for l in range(2, S):
        P.append(((2 * l - 1) * cos * P[l - 1] - (l - 1) * P[l - 2]) / l)
    return torch.stack([math.sqrt((2 * l + 1) / (4 * math.pi)) * P[l] for l in range(S)], -1)


def noise_level_encoding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """Sinusoidal encoding of the diffusion time t in (0, 1], position = 1000 t: [B, dim] = [sin | cos]."""
    half = dim // 2
    div = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000.0) / half))
    arg = (t.to(torch.float32) * 1000.0)[:, None] * div[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], 1)


def ssilu(x):
    return torch.nn.functional.silu(x) * ACT_SCALE


# --------------------------------------------------------------------------------------------------------------------------
# the denoiser
# --------------------------------------------------------------------------------------------------------------------------
def _dense(P, name, x, act=False):
    y = x @ P[name + ".weight"].t()
    return ssilu(y) if act else y


def _residual_stack(P, prefix, n, x):
    for k in range(n):
        y = _dense(P, f"{prefix}.{k}.1", _dense(P, f"{prefix}.{k}.0", x, True), True)
        x = (x + y) * INV_SQRT_2
    return x


def _segment_sum(x, dst, N):
    return torch.zeros(N, x.shape[1], dtype=x.dtype).index_add(0, dst, x)


def triplet_sum(xd, V, g, S):
    """sum_k[e][l][j] = sum over the OTHER in-edges k of e's target atom of Y_l(V_e . V_k) xd[k][j]: [E, S, emb_trip]."""
    E = xd.shape[0]
    out = torch.zeros(E, S, xd.shape[1], dtype=xd.dtype)
    rowptr = g["rowptr"].tolist()
    for a in range(len(rowptr) - 1):
        lo, hi = rowptr[a], rowptr[a + 1]
        if hi - lo < 2:
            continue
        Va = V[lo:hi]
        cos = (Va @ Va.t()).clamp(-1.0, 1.0)                       # [deg, deg]
        Y = legendre_sph(cos, S)                                   # [deg(e), deg(k), S]
        Y = Y * (1.0 - torch.eye(hi - lo))[:, :, None]             # k != e
        out[lo:hi] = torch.einsum("ekl,kj->elj", Y, xd[lo:hi])
    return out


def gemnet_forward(P: Dict[str, torch.Tensor], hp: GemNetHParams, frac, cell, atomic_numbers, num_atoms, t, graph=None, taps=None):
    """GemNet-T-shaped denoiser: returns dict(pos [N,3] (fractional, = score x std), cell [B,3,3] (symmetric), atomic_numbers
    [N,101] logits).  `taps` (dict) receives intermediates for the layer-by-layer parity tests."""
    g = graph if graph is not None else build_graph(frac.detach(), cell.detach(), num_atoms, hp)
    B, N = int(num_atoms.shape[0]), int(frac.shape[0])
    src, dst, swap, n2g, eg = g["src"], g["dst"], g["swap"], g["node2graph"], g["edge_graph"]
    E = int(src.shape[0])
    A, S, Cb, Tr = hp.emb_atom, hp.num_spherical, hp.emb_cbf, hp.emb_trip
    D, V = g["D"], g["V"]
    rbf = radial_basis(D, hp)
    z = noise_level_encoding(t, A)
    h = P["atom_emb.weight"][atomic_numbers - 1]
    h = torch.cat([h, z[n2g]], 1) @ P["atom_latent_emb.weight"].t() + P["atom_latent_emb.bias"]
    m = ssilu(torch.cat([h[src], h[dst], rbf], 1) @ P["edge_emb.weight"].t())
    rbf3 = rbf @ P["mlp_rbf3.weight"].t()
    cbfW = (rbf @ P["mlp_cbf3.weight"].t()).reshape(E, S, Cb)      # [e][l][i]
    rbf_h = rbf @ P["mlp_rbf_h.weight"].t()
    rbf_out = rbf @ P["mlp_rbf_out.weight"].t()
    if taps is not None:
        taps.update(rbf=rbf, h0=h, m0=m, D=D, V=V)

    def sf(name):   # ScalingFactor: a constant (detached: it is a buffer upstream, never a trained parameter)
        return P[name + ".scale_factor"].detach().reshape(())

    def out_block(i, m):
        p = f"out_blocks.{i}"
        # energy path (atom level): E_t is computed by the upstream denoiser and discarded by the score model -- restated for the parameter
        # list's sake and returned under "energy"; the device path does not evaluate it (no output depends on it)
        xE = _segment_sum(m * (rbf_out @ P[f"{p}.dense_rbf.weight"].t()), dst, N) * sf(f"{p}.scale_sum")
        xE = _residual_stack(P, f"{p}.seq_energy.res", hp.num_atom, _dense(P, f"{p}.seq_energy.dense", xE, True))
        En = xE @ P[f"{p}.out_energy.weight"].t()
        # force path (edge level)
        xF = _residual_stack(P, f"{p}.res_F", hp.num_atom, _dense(P, f"{p}.dense_F", m, True))
        F = ((xF * (rbf_out @ P[f"{p}.rbf_F.weight"].t())) * sf(f"{p}.scale_rbf_F")) @ P[f"{p}.out_F.weight"].t()
        xS = _dense(P, f"{p}.dense_S", m, True)
        Sc = (xS * (rbf_out @ P[f"{p}.rbf_S.weight"].t())) @ P[f"{p}.out_S.weight"].t()
        return F, Sc, En

    F, Sc, En = out_block(0, m)
    for i in range(hp.num_blocks):
        p = f"int_blocks.{i}"
        x_ca = _dense(P, f"{p}.dense_ca", m, True)
        x_ba = _dense(P, f"{p}.dense_ba", m, True) * (rbf3 @ P[f"{p}.mlp_rbf.weight"].t()) * sf(f"{p}.scale_rbf")
        xd = x_ba @ P[f"{p}.down_projection.weight"].t()                                   # [E, Tr]
        sk = triplet_sum(xd, V, g, S)                                                      # [E, S, Tr]
        tm = torch.einsum("eli,elj->eij", cbfW, sk).reshape(E, Cb * Tr)                    # [e][i][j]
        x3 = (tm @ P[f"{p}.bilinear.weight"].t()) * sf(f"{p}.scale_cbf_sum")               # [E, Bl]
        if taps is not None:
            taps[f"x3_{i}"] = x3
        x3 = (_dense(P, f"{p}.up_projection_ca", x3, True) + _dense(P, f"{p}.up_projection_ac", x3, True)[swap]) * INV_SQRT_2
        x = (x_ca + x3) * INV_SQRT_2
        x = _residual_stack(P, f"{p}.before_skip", hp.num_before_skip, x)
        m = (m + x) * INV_SQRT_2
        m = _residual_stack(P, f"{p}.after_skip", hp.num_after_skip, m)
        h2 = _segment_sum(m * (rbf_h @ P[f"{p}.atom_update.rbf.weight"].t()), dst, N) * sf(f"{p}.atom_update.scale_sum")
        h2 = _residual_stack(P, f"{p}.atom_update.res", hp.num_atom, _dense(P, f"{p}.atom_update.dense", h2, True))
        h = (h + h2) * INV_SQRT_2
        m2 = ssilu(torch.cat([h[src], h[dst], m], 1) @ P[f"{p}.concat.weight"].t())
        m2 = _residual_stack(P, f"{p}.residual_m", hp.num_concat, m2)
        m = (m + m2) * INV_SQRT_2
        Fi, Si, Ei = out_block(i + 1, m)
        F, Sc, En = F + Fi, Sc + Si, En + Ei
        if taps is not None:
            taps[f"h{i + 1}"], taps[f"m{i + 1}"] = h, m
    force = _segment_sum(F * V, dst, N)                                                    # [N,3] cartesian
    pos = torch.einsum("ni,nij->nj", force, torch.linalg.inv(cell)[n2g])                   # fractional
    outer = V[:, :, None] * V[:, None, :]
    cnt = torch.zeros(B).index_add(0, eg, torch.ones(E)).clamp(min=1.0)
    stress = torch.zeros(B, 3, 3).index_add(0, eg, Sc[:, :, None] * outer) / cnt[:, None, None]
    logits = h @ P["fc_atom.weight"].t() + P["fc_atom.bias"]
    return dict(pos=pos, cell=stress, atomic_numbers=logits, energy=En)


# --------------------------------------------------------------------------------------------------------------------------
# corruptions (wrapped VE-SDE on positions, VP-SDE with a density-dependent limit on the cell, D3PM absorbing on the types)
# --------------------------------------------------------------------------------------------------------------------------
@dataclass
class Corruption:
    sigma_min: float = 0.005
    sigma_max: float = 5.0
    beta_min: float = 0.1
    beta_max: float = 20.0
    limit_density: float = 0.05771451654022283
    limit_var_scale: float = 0.25
    d3pm_steps: int = 1000
    T: float = 1.0

    def pos_std(self, t, num_atoms):
        """sigma(t) n^(-1/3): the VE std in FRACTIONAL units adjusted for the cell size growing with the atom count."""
        sig = self.sigma_min ** (1 - t) * self.sigma_max ** t
        return sig * num_atoms.to(torch.float32) ** (-1.0 / 3.0)

    def cell_alpha(self, t):
        return torch.exp(-0.5 * (t * self.beta_min + 0.5 * t * t * (self.beta_max - self.beta_min)))

    def cell_limit(self, num_atoms):
        n = num_atoms.to(torch.float32)
        return (n / self.limit_density) ** (1.0 / 3.0), math.sqrt(self.limit_var_scale) * n ** (1.0 / 3.0)   # mean scale, std

    def beta(self, t):
        return self.beta_min + t * (self.beta_max - self.beta_min)


def symmetric_noise(G):
    """[B,3,3] i.i.d. normals -> symmetric with unit variance per entry: diagonal kept, off-diagonal (G_ij + G_ji) / sqrt(2)."""
    S = (G + G.transpose(1, 2)) * INV_SQRT_2
    eye = torch.eye(3)[None]
    return S * (1 - eye) + G * eye


def d_log_p_wrapped_normal(x, sigma, N=10):
    """Score of the wrapped normal (period 1), 21 images -- the same helper DiffCSP uses (models/diffcsp/scheduler.py:39-43)."""
    num = torch.zeros_like(x)
    den = torch.zeros_like(x)
    for i in range(-N, N + 1):
        e = torch.exp(-(x + i) ** 2 / 2 / sigma ** 2)
        num = num + (x + i) / sigma ** 2 * e
        den = den + e
    return -num / den          # NB: d/dx log p = -sum (x+i)/s^2 e / sum e


def time_grid(corr: Corruption, timestep: int, n: int = 1000):
    """MatterGenModule.add_noise (pl_module.py:57-62): linspace(T_max, 1/N, N)[timestep]."""
    return float(torch.linspace(corr.T, 1.0 / n, n)[timestep])


def sample_marginal(corr: Corruption, batch: dict, t: torch.Tensor, noise: dict):
    """corruption.sample_marginal (pl_module.py:68).  batch: pos, cell, atomic_numbers, num_atoms; t [B]; noise: pos [N,3] normal,
    cell [B,3,3] normal, types [N] uniform.  Returns the noisy batch and what the loss needs."""
    na = batch["num_atoms"]
    n2g = torch.repeat_interleave(torch.arange(len(na)), na)
    std = corr.pos_std(t, na)[n2g][:, None]
    delta = std * noise["pos"]
    pos = (batch["pos"] + delta) % 1.0
    alpha = corr.cell_alpha(t)[:, None, None]
    mu, kap = corr.cell_limit(na)
    eps = symmetric_noise(noise["cell"])
    cell = alpha * batch["cell"] + (1 - alpha) * mu[:, None, None] * torch.eye(3)[None] + torch.sqrt(1 - alpha * alpha) * kap[:, None, None] * eps
    tau = torch.clamp(torch.ceil(t * corr.d3pm_steps - 1e-6), 1, corr.d3pm_steps)[n2g]
    masked = noise["types"] < tau / corr.d3pm_steps
    types = torch.where(masked, torch.full_like(batch["atomic_numbers"], MASK), batch["atomic_numbers"])
    noisy = dict(pos=pos, cell=cell, atomic_numbers=types, num_atoms=na)
    aux = dict(delta=delta, std=std, eps=eps, tau=tau, masked=masked, node2graph=n2g)
    return noisy, aux


def _scatter_mean(v, idx, B):
    s = torch.zeros(B, dtype=v.dtype).index_add(0, idx, v)
    c = torch.zeros(B, dtype=v.dtype).index_add(0, idx, torch.ones_like(v))
    return s / c.clamp(min=1)


WEIGHTS = dict(atomic_numbers=1.0, cell=1.0, pos=0.1)     # loss.py:22-26
D3PM_LAMBDA = 0.01                                       # loss.py:15


def sample_loss(corr: Corruption, batch, aux, pred):
    """SampleLoss.__call__ (loss.py:36-78): per-crystal sum_field w_field * loss_field.
      pos:  mean over atoms and dims of (pred_pos - std * score_wn(delta; std))^2     (denoising score matching, wrapped normal)
      cell: mean over the 9 entries of (pred_cell + eps)^2                            (the model estimates score x std = -eps)
      atomic_numbers: mean over atoms of [masked] (1/tau) (-log p(x0)) + 0.01 (-log p(x0)), p = softmax over the 100 ELEMENT logits
                      (D3PM absorbing-state hybrid loss: KL(q(x_{tau-1} | x_tau, x0) || p_theta) + lambda CE)."""
    B = len(batch["num_atoms"])
    n2g = aux["node2graph"]
    target = aux["std"] * d_log_p_wrapped_normal(aux["delta"], aux["std"])
    l_pos = _scatter_mean(((pred["pos"] - target) ** 2).mean(1), n2g, B)
    l_cell = ((pred["cell"] + aux["eps"]) ** 2).mean(dim=(1, 2))
    logp = torch.log_softmax(pred["atomic_numbers"][:, :NUM_CLASSES - 1], dim=1)
    nll = -logp.gather(1, (batch["atomic_numbers"] - 1)[:, None])[:, 0]
    l_types = _scatter_mean(aux["masked"].to(nll.dtype) * nll / aux["tau"] + D3PM_LAMBDA * nll, n2g, B)
    fields = dict(pos=l_pos, cell=l_cell, atomic_numbers=l_types)
    return weighted_field_sum(fields), fields


def weighted_field_sum(fields, weights=None):
    """SampleLoss.__call__'s aggregation (loss.py:71-73): stack of w_field * loss_field over the fields, summed -> [B].  Pinned by the reference-generated
    fixture tests/golden/g12_mattergen_adapter.npz (the per-field losses themselves are upstream arithmetic and stay unpinned)."""
    w = WEIGHTS if weights is None else weights
    return torch.stack([w[k] * v for k, v in fields.items()], dim=0).sum(0)


def calc_kl_reg(agent_pred, prior_pred, node2graph, B):
    """MatterGenModule.calc_kl_reg (pl_module.py:83-102)."""
    k0 = ((agent_pred["cell"] - prior_pred["cell"].detach()) ** 2).mean(dim=(1, 2))
    k1 = _scatter_mean(((agent_pred["pos"] - prior_pred["pos"].detach()) ** 2).mean(1), node2graph, B)
    k2 = _scatter_mean(((agent_pred["atomic_numbers"] - prior_pred["atomic_numbers"].detach()) ** 2).mean(1), node2graph, B)
    return k0 + k1 + k2


# --------------------------------------------------------------------------------------------------------------------------
# predictor-corrector sampler (N steps on linspace(T, eps_t, N); one Langevin corrector step + one ancestral predictor step per
# time, two denoiser evaluations per step; returns the final `mean` batch like models/mattergen/sample.py:49-50)
# --------------------------------------------------------------------------------------------------------------------------
SNR_POS, SNR_CELL = 0.4, 0.2


def _per_crystal_norm(v, idx, B):
    """sqrt(sum over a crystal's entries of v^2)."""
    return torch.sqrt(torch.zeros(B, dtype=v.dtype).index_add(0, idx, (v * v).reshape(v.shape[0], -1).sum(1)))


def pc_sample(P, hp: GemNetHParams, corr: Corruption, num_atoms: torch.Tensor, noise: dict, n_steps: int = 1000, eps_t: float = 1e-3,
              t_stop_index: Optional[int] = None, record=None, start_index: int = 0, state: Optional[dict] = None):
    """noise: init_pos [N,3] uniform, init_cell [B,3,3] normal, and per step index i: corr_pos[i], corr_cell[i], pred_pos[i],
    pred_cell[i] normals, pred_u1[i], pred_u2[i] [N] uniforms.  Returns (sample, mean) dicts."""
    B, N = len(num_atoms), int(num_atoms.sum())
    n2g = torch.repeat_interleave(torch.arange(B), num_atoms)
    eye = torch.eye(3)[None]
    mu, kap = corr.cell_limit(num_atoms)
    types = torch.full((N,), MASK, dtype=torch.long)
    if state is None:
        pos = noise["init_pos"] % 1.0
        cell = mu[:, None, None] * eye + kap[:, None, None] * symmetric_noise(noise["init_cell"])
    if state is not None:   # resume a chain at grid point `start_index` from a given state (teacher-forced tests)
        pos, cell, types = state["pos"] % 1.0, state["cell"], state["atomic_numbers"]
    ts = torch.linspace(corr.T, eps_t, n_steps)
    dt = float(ts[0] - ts[1]) if n_steps > 1 else corr.T - eps_t
    mean = dict(pos=pos, cell=cell, atomic_numbers=types)
    last = n_steps if t_stop_index is None else t_stop_index
    for i in range(start_index, last):
        t = torch.full((B,), float(ts[i]))
        std_p = corr.pos_std(t, num_atoms)
        alpha = corr.cell_alpha(t)
        std_c = torch.sqrt(1 - alpha * alpha) * kap
        # ---- corrector (Langevin, per-crystal signal-to-noise step size) ----
        out = gemnet_forward(P, hp, pos, cell, types, num_atoms, t)
        sc_p = out["pos"] / std_p[n2g][:, None]
        z = noise["corr_pos"][i]
        step = 2.0 * (SNR_POS * _per_crystal_norm(z, n2g, B) / _per_crystal_norm(sc_p, n2g, B).clamp(min=1e-12)) ** 2
        step = step.clamp(max=1e6)[n2g][:, None]
        pos = (pos + step * sc_p + torch.sqrt(2 * step) * z) % 1.0
        sc_c = out["cell"] / std_c[:, None, None]
        zc = symmetric_noise(noise["corr_cell"][i])
        ar = torch.arange(B)
        stepc = 2.0 * (SNR_CELL * _per_crystal_norm(zc, ar, B) / _per_crystal_norm(sc_c, ar, B).clamp(min=1e-12)) ** 2
        stepc = stepc.clamp(max=1e6)[:, None, None]
        cell = cell + stepc * sc_c + torch.sqrt(2 * stepc) * zc
        # ---- predictor (ancestral) ----
        out = gemnet_forward(P, hp, pos, cell, types, num_atoms, t)
        t_next = torch.clamp(t - dt, min=0.0)
        std_n = corr.pos_std(t_next, num_atoms) if i + 1 < n_steps else torch.zeros(B)
        var_d = (std_p ** 2 - std_n ** 2)[n2g][:, None]
        sc_p = out["pos"] / std_p[n2g][:, None]
        pos_mean = pos + var_d * sc_p
        pos = (pos_mean + torch.sqrt((std_n ** 2)[n2g][:, None] * var_d / (std_p ** 2)[n2g][:, None]) * noise["pred_pos"][i]) % 1.0
        bd = (corr.beta(t) * dt)[:, None, None]
        sc_c = out["cell"] / std_c[:, None, None]
        cell_mean = cell + 0.5 * bd * (cell - mu[:, None, None] * eye) + bd * (kap ** 2)[:, None, None] * sc_c
        cell = cell_mean + torch.sqrt(bd) * kap[:, None, None] * symmetric_noise(noise["pred_cell"][i])
        tau = torch.clamp(torch.ceil(t * corr.d3pm_steps - 1e-6), 1, corr.d3pm_steps)[n2g]
        p = torch.softmax(out["atomic_numbers"][:, :NUM_CLASSES - 1], dim=1)
        cdf = torch.cumsum(p, 1)
        draw = (noise["pred_u2"][i][:, None] >= cdf).sum(1).clamp(max=NUM_CLASSES - 2) + 1
        unmask = (types == MASK) & (noise["pred_u1"][i] < 1.0 / tau)
        types = torch.where(unmask, draw, types)
        mean = dict(pos=pos_mean % 1.0, cell=cell_mean, atomic_numbers=types)
        if record is not None:
            record.append(dict(pos=pos.clone(), cell=cell.clone(), atomic_numbers=types.clone(), mean_pos=mean["pos"].clone(),
                               mean_cell=mean["cell"].clone()))
    return dict(pos=pos, cell=cell, atomic_numbers=types, num_atoms=num_atoms), dict(mean, num_atoms=num_atoms)
