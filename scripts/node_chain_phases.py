"""Phase clock of the node-chain launch (csrc/node_chain.hip): where a workgroup's time goes, alone on the chip.
usage (GPU box): python scripts/node_chain_phases.py [crystals]"""
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
torch.manual_seed(0)
net = CSPNet(hidden_dim=H, num_layers=L, num_freqs=F, latent_dim=256, ln=True, smooth=True, pred_type=True, device="cuda")
g = torch.Generator().manual_seed(1)
N = B * n
t_emb = O.time_embedding(torch.full((B,), 500), 256).cuda()
at, fr = torch.randn(N, 100, generator=g).cuda(), torch.rand(N, 3, generator=g).cuda()
lat = (4 * torch.eye(3) + torch.randn(B, 3, 3, generator=g)).cuda()
bt = net.make_batch([n] * B)
torch.set_grad_enabled(False)   # inference forwards (the training forward keeps the seven-launch form and its tape)
for _ in range(3):
    net(t_emb, at, fr, lat, None, batch=bt)
nwg = (N + 31) // 32
clk = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
lib.mi_debug_node_chain_clock(C.c_void_p(clk.data_ptr()))
net(t_emb, at, fr, lat, None, batch=bt)   # the buffer keeps the LAST launch that stamps each slot: the final chain (phase A + LayerNorm only) overwrites 0..6
torch.cuda.synchronize()
lib.mi_debug_node_chain_clock(None)
c = clk.cpu().numpy().reshape(nwg, 16)
names = ["A1 agg", "A2 gemm", "A3 epi", "A4 gemm", "A5 epi", "LN", "B pass0", "B pass1", "B pass2", "B tail"]
d = np.diff(c[:, :11], axis=1)
print(f"{B} crystals, {nwg} workgroups; s_memtime ticks (100 MHz => 10 ns each), mean / max over workgroups")
for k, nm in enumerate(names):
    print(f"  {nm:8s} {d[:, k].mean():8.1f} {d[:, k].max():8d}")
print("  total   ", (c[:, 10] - c[:, 0]).mean(), " span of the launch:", c[:, :11].max() - c[:, 0].min())
