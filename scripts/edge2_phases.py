"""Phase clock of the 128-row x 512-column second edge GEMM (csrc/edge_stage.hip), alone on the chip.
usage (GPU box): python scripts/edge2_phases.py [crystals] [variant 1|2]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from matinvent_amd import _lib  # noqa: E402
from matinvent_amd.cspnet import CSPNet  # noqa: E402
from oracle import diffcsp_oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
var = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, H, L, F = 20, 512, 6, 128
lib = _lib.load()
lib.mi_debug_set_edge2_fused(var)
torch.manual_seed(0)
net = CSPNet(hidden_dim=H, num_layers=L, num_freqs=F, latent_dim=256, ln=True, smooth=True, pred_type=True, device="cuda")
g = torch.Generator().manual_seed(1)
N = B * n
t_emb = O.time_embedding(torch.full((B,), 500), 256).cuda()
at, fr = torch.randn(N, 100, generator=g).cuda(), torch.rand(N, 3, generator=g).cuda()
lat = (4 * torch.eye(3) + torch.randn(B, 3, 3, generator=g)).cuda()
bt = net.make_batch([n] * B)
torch.set_grad_enabled(False)
for _ in range(3):
    net(t_emb, at, fr, lat, None, batch=bt)
nt = (B * n * n + 127) // 128 * (2 if var == 1 else 1)   # (form B: two column halves per row tile)
clk = torch.zeros(nt * 8, dtype=torch.int64, device="cuda")
lib.mi_debug_edge2_clock(C.c_void_p(clk.data_ptr()))
net(t_emb, at, fr, lat, None, batch=bt)
torch.cuda.synchronize()
lib.mi_debug_edge2_clock(None)
c = clk.cpu().numpy().reshape(nt, 8)
d = np.diff(c[:, :4], axis=1)
print(f"{B} crystals, {nt} row tiles, variant {var}; s_memtime ticks, mean / max over tiles (last layer's launch)")
for k, nm in enumerate(["first chunk + ring + tables", "main loop (4 chunks x 8 k-steps)", "epilogue"]):
    print(f"  {nm:34s} {d[:, k].mean():9.1f} {d[:, k].max():9d}")
print("  span of the launch:", c[:, 3].max() - c[:, 0].min(), " mean tile:", (c[:, 3] - c[:, 0]).mean())
if var == 1:   # stamps 4 / 5: s_memrealtime (100 MHz) at the tile's first and last stamp -> the shader clock the s_memtime stamps ran at
    rt = (c[:, 5] - c[:, 4]).astype(np.float64)
    ok = rt > 0
    print(f"  shader clock during the launch: {np.median((c[ok, 3] - c[ok, 0]) / rt[ok]) * 0.1:.2f} GHz (median over tiles; realtime span of the launch "
          f"{(c[ok, 5].max() - c[ok, 4].min()) * 0.01:.1f} us)")
if var == 1:   # occupancy over the launch from the realtime stamps (10 ns ticks): how many workgroups are alive, decile by decile
    t0, t1 = c[ok, 4].min(), c[ok, 5].max()
    edges = np.linspace(t0, t1, 11)
    alive = [np.mean([np.sum((c[ok, 4] <= x) & (c[ok, 5] > x)) for x in np.linspace(edges[i], edges[i + 1], 20, endpoint=False)]) for i in range(10)]
    print("  workgroups alive, by decile of the launch:", " ".join(f"{a:.0f}" for a in alive), f"  (512 slots; workgroup-time / (span x 512) = "
          f"{np.sum(c[ok, 5] - c[ok, 4]) / ((t1 - t0) * 512.0):.2f})")
    life = (c[ok, 5] - c[ok, 4]) * 0.01
    print(f"  workgroup life: median {np.median(life):.1f} us, 10 % / 90 %: {np.percentile(life, 10):.1f} / {np.percentile(life, 90):.1f} us")
# the first edge GEMM (pair mode)
npairs = B * n * (n - 1) // 2
nt1 = ((npairs + 127) // 128 + 7) // 8 * 8 * 4
clk1 = torch.zeros(nt1 * 8, dtype=torch.int64, device="cuda")
lib.mi_debug_edge1_clock(C.c_void_p(clk1.data_ptr()))
net(t_emb, at, fr, lat, None, batch=bt)
torch.cuda.synchronize()
lib.mi_debug_edge1_clock(None)
c1 = clk1.cpu().numpy().reshape(nt1, 8)
c1 = c1[c1[:, 3] > 0]
if len(c1) == 0:
    print("first edge GEMM: on the plane GEMM (mi_debug_set_edge1_fused(1) selects the register-tile form with this clock)")
    sys.exit(0)
d1 = np.diff(c1[:, :4], axis=1)
print(f"first edge GEMM: {len(c1)} workgroups (128 pairs x 128 columns)")
for k, nm in enumerate(["first k-tile + ring", "main loop (24 k-tiles)", "epilogue (both edges of a pair)"]):
    print(f"  {nm:34s} {d1[:, k].mean():9.1f} {d1[:, k].max():9d}")
print("  span of the launch:", c1[:, 3].max() - c1[:, 0].min(), " mean workgroup:", (c1[:, 3] - c1[:, 0]).mean())
