"""Phase clock of one gemm_rt launch inside a forward of the MatterGen-shaped network at the benchmark size (256 crystals x 20 atoms, ~250 k
edges): s_memtime ticks per workgroup for set-up / main loop / epilogue, the shader clock they ran at, and how many workgroups were alive over the
launch.  usage (GPU box): python scripts/rt_phases.py [ext 0|1] [skip]"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import mattergen_oracle as M
from matinvent_amd import _lib
from matinvent_amd.mattergen import MatterGenModule
lib = _lib.load()
ext = int(sys.argv[1]) if len(sys.argv) > 1 else 0
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hp = M.GemNetHParams()
P = M.init_params(hp, seed=0, head_scale=20.0)
m = MatterGenModule(gemnet={}); m.decoder.load_state_dict(P, strict=True)
n, B = 20, 256
g = torch.Generator().manual_seed(3)
mu = (n / 0.05771451654022283) ** (1 / 3)
cell = (mu * torch.eye(3)[None].repeat(B, 1, 1) + 0.3 * M.symmetric_noise(torch.randn(B, 3, 3, generator=g))).cuda()
na = torch.full((B,), n, dtype=torch.long)
frac, a, t = torch.rand(B * n, 3, generator=g).cuda(), torch.randint(1, 101, (B * n,), generator=g).cuda(), (0.1 + 0.8 * torch.rand(B, generator=g)).cuda()
gb = m.decoder.make_batch(na)
with torch.no_grad():
    m.decoder(frac, cell, a, t, gb); m.decoder(frac, cell, a, t, gb)
    torch.cuda.synchronize()
    nwg = 2 * ((300000 + 127) // 128 + 8)
    clk = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    lib.mi_debug_rt_clock(C.c_void_p(clk.data_ptr()), ext, skip)
    m.decoder(frac, cell, a, t, gb)
    torch.cuda.synchronize()
    lib.mi_debug_rt_clock(None, -1, 0)
c = clk.cpu().numpy().reshape(nwg, 8)
c = c[c[:, 3] > 0]
if c[:, 6].max() > 0:   # the persistent form: [0..3] = the FIRST tile's stamps, [6] = tiles done, [7] = s_memtime at exit
    tiles = c[:, 6].astype(np.float64)
    whole = (c[:, 7] - c[:, 0]).astype(np.float64)
    print(f"persistent grid: {len(c)} workgroups, {int(tiles.sum())} tiles, {tiles.mean():.2f} per workgroup; cycles per tile (whole workgroup / its tiles): mean {np.mean(whole / tiles):.0f}")
d = np.diff(c[:, :4], axis=1).astype(np.float64)
rt = (c[:, 5] - c[:, 4]).astype(np.float64)
ok = rt > 0
ghz = np.median((c[ok, 3] - c[ok, 0]) / rt[ok]) * 0.1
span_us = (c[:, 5].max() - c[:, 4].min()) * 0.01
print(f"gemm_rt<ext={ext}> launch #{skip}: {len(c)} workgroups, launch span {span_us:.1f} us, shader clock {ghz:.2f} GHz (s_memtime ticks at that clock)")
for k, nm in enumerate(["set-up (first k-tile in LDS, weight ring)", "main loop", "epilogue"]):
    print(f"  {nm:44s} mean {d[:, k].mean():9.0f}  median {np.median(d[:, k]):9.0f}  max {d[:, k].max():9.0f}")
tot = (c[:, 3] - c[:, 0]).astype(np.float64)
print(f"  workgroup total: mean {tot.mean():.0f} ticks = {tot.mean() / ghz / 1e3:.1f} us; MFMA floor of a workgroup alone on its SIMDs: 768 MFMAs x 32 = 24.6 k cycles, of two sharing them 49 k")
t0, t1 = c[:, 4].min(), c[:, 5].max()
edges = np.linspace(t0, t1, 11)
alive = [int(((c[:, 4] < edges[i + 1]) & (c[:, 5] > edges[i])).sum()) for i in range(10)]
print("  workgroups alive per decile of the launch:", alive)
