"""Phase clock of the one-launch edge stage (csrc/edge_fused.hip), alone on the chip.  usage (GPU box): python scripts/edge_fused_phases.py [crystals]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from matinvent_amd import _lib  # noqa: E402
from matinvent_amd.cspnet import CSPNet  # noqa: E402
from oracle import diffcsp_oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, H, L, F = 20, 512, 6, 128
lib = _lib.load()
lib.mi_debug_set_edge_fused(1)
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
npt = (B * n * (n - 1) // 2 + 63) // 64
nt = npt + (N + 127) // 128
clk = torch.zeros(nt * 16, dtype=torch.int64, device="cuda")
lib.mi_debug_edge_fused_clock(C.c_void_p(clk.data_ptr()))
net(t_emb, at, fr, lat, None, batch=bt)
torch.cuda.synchronize()
lib.mi_debug_edge_fused_clock(None)
c = clk.cpu().numpy().reshape(nt, 16)[:npt]
print(f"{B} crystals, {npt} pair tiles; s_memtime ticks, mean over tiles (last layer's launch)")
print(f"  entry -> first product of chunk 0 done   {np.mean(c[:, 1] - c[:, 0]):9.1f}")
for ch in range(4):
    b0 = c[:, 3 * ch] if ch else c[:, 0]
    print(f"  chunk {ch}: first product {np.mean(c[:, 1 + 3 * ch] - b0):9.1f}   pair epilogue {np.mean(c[:, 2 + 3 * ch] - c[:, 1 + 3 * ch]):9.1f}   "
          f"second product {np.mean(c[:, 3 + 3 * ch] - c[:, 2 + 3 * ch]):9.1f}")
print(f"  final epilogue {np.mean(c[:, 13] - c[:, 12]):9.1f}   whole tile {np.mean(c[:, 13] - c[:, 0]):9.1f}")
