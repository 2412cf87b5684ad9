#!/usr/bin/env python
"""Parameter-gradient error of the hand-written backward against the oracle's autograd evaluated in fp64, at a size where the
large-problem kernels run (B=96 x 20 atoms, H=512, L=2, F=128) -- the numbers DESIGN.md quotes for the fp16 / bf16 operand formats
of the backward products.  `--bf16` selects the three-plane bf16 kernels (mi_debug_set_tn128(3 + 32 + 64))."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import diffcsp_oracle as O  # noqa: E402
from tests.gpu_util import make_module  # noqa: E402


def main():
    H, L, F = 512, 2, 128
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=6)
    gen = torch.Generator().manual_seed(12)
    m = make_module(H, L, F, 20, P)
    if "--bf16" in sys.argv:
        from matinvent_amd import _lib
        _lib.check(_lib.load().mi_debug_set_tn128(3 + 32 + 64))
    B, n = 96, 20
    na = torch.full((B,), n, dtype=torch.long)
    N = B * n
    n2g = torch.repeat_interleave(torch.arange(B), na)
    t_emb = O.time_embedding(torch.full((B,), 7), 256)
    at, fr = torch.randn(N, 100, generator=gen), torch.rand(N, 3, generator=gen)
    lat = 4 * torch.eye(3) + torch.randn(B, 3, 3, generator=gen)
    ul, ux, ut = torch.randn(B, 3, 3, generator=gen), torch.randn(N, 3, generator=gen), torch.randn(N, 100, generator=gen)

    def oracle(dtype):
        Pg = {k: v.to(dtype).clone().requires_grad_(True) for k, v in P.items()}
        ol, ox, ot = O.cspnet_forward(Pg, hp, t_emb.to(dtype), at.to(dtype), fr.to(dtype), lat.to(dtype), na, n2g)
        ((ol * ul.to(dtype)).sum() + (ox * ux.to(dtype)).sum() + (ot * ut.to(dtype)).sum()).backward()
        return {k: v.grad.double() for k, v in Pg.items()}
    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    pl, px, pt = m.decoder(t_emb.cuda(), at.cuda(), fr.cuda(), lat.cuda(), na)
    ((pl * ul.cuda()).sum() + (px * ux.cuda()).sum() + (pt * ut.cuda()).sum()).backward()
    th = m.decoder.theta.grad.double().cpu()
    worst = {}
    for k, (o, cnt, shape) in m.decoder.layout.items():
        ref = g64["decoder." + k]
        sc = float(ref.abs().max()) or 1.0
        e_hip = float((th[o:o + cnt].view(shape) - ref).abs().max()) / sc
        e_f32 = float((g32["decoder." + k] - ref).abs().max()) / sc
        grp = k.split(".")[-2] + "." + k.split(".")[-1] if "csp_layer" in k else k
        w = worst.setdefault(grp, [0.0, 0.0])
        w[0], w[1] = max(w[0], e_hip), max(w[1], e_f32)
    print("max |grad - fp64 grad| / max|fp64 grad| per tensor class:   HIP      fp32 oracle")
    for k, (a, b) in worst.items():
        print(f"  {k:32s} {a:9.2e} {b:9.2e}")
    print(f"  {'ALL':32s} {max(a for a, _ in worst.values()):9.2e} {max(b for _, b in worst.values()):9.2e}")


if __name__ == "__main__":
    main()
