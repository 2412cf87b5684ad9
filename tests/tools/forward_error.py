"""Deviation of the HIP forward from the CPU oracle (fp32) and from the oracle evaluated in fp64, at the benchmark shape
(B = 256 x 20 atoms, H = 512, L = 6, F = 128): max |difference| / max(1, max |reference|) per output.  The fp64 column separates
the arithmetic error of the plane-set GEMMs from the fp32 oracle's own rounding.  Usage (GPU box): python tests/tools/forward_error.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import diffcsp_oracle as O
from tests.gpu_util import load_decoder
from matinvent_amd.cspnet import CSPNet

B, n, H, L, F = 256, 20, 512, 6, 128
hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
P = O.init_params(hp, seed=4)
net = CSPNet(hidden_dim=H, num_layers=L, num_freqs=F, latent_dim=256, ln=True, smooth=True, pred_type=True, device="cuda")
net.set_edge_mode("gemm")
load_decoder(net, P)
g = torch.Generator().manual_seed(17)
N = B * n
na = torch.full((B,), n, dtype=torch.long)
n2g = torch.repeat_interleave(torch.arange(B), na)
t_emb = O.time_embedding(torch.full((B,), 640), 256)
at = torch.randn(N, 100, generator=g)
fr = torch.rand(N, 3, generator=g)
for lat_scale, label in ((1.0, "cells ~4 A"), (30.0, "cells ~120 A (early-chain magnitudes)")):
    lat = lat_scale * (4 * torch.eye(3) + torch.randn(B, 3, 3, generator=g))
    with torch.no_grad():
        ref32 = O.cspnet_forward(P, hp, t_emb, at, fr, lat, na, n2g)
        P64 = {k: (v.double() if torch.is_floating_point(v) else v) for k, v in P.items()}
        ref64 = O.cspnet_forward(P64, hp, t_emb.double(), at.double(), fr.double(), lat.double(), na, n2g)
    out = net(t_emb.cuda(), at.cuda(), fr.cuda(), lat.cuda(), None, batch=net.make_batch([n] * B))
    for a, r32, r64, w in zip(out, ref32, ref64, ("pred_l", "pred_x", "pred_t")):
        a = a.detach().cpu().double()
        s = max(1.0, float(r64.abs().max()))
        print(f"{label:40s} {w}: vs fp32 oracle {float((a - r32.double()).abs().max()) / s:.2e}   vs fp64 {float((a - r64).abs().max()) / s:.2e}"
              f"   (fp32 oracle vs fp64 {float((r32.double() - r64).abs().max()) / s:.2e})")
