"""Micro-benchmark of the GEMM kernels through mi_debug_gemm (kind 0 = f32-input MFMA, 2 = pre-split tile-blocked planes on the
128x128 two-workgroups-per-CU kernel, 3 = the 256x128 double-buffered kernel): time and max relative error against fp64.  GPU only."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matinvent_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
for (M, N, K) in [(102400, 512, 768), (102400, 512, 512), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5
    ref = (A.double() @ W.double().t())
    Cm = torch.empty(M, N, device='cuda')
    for kind, name in [(0, 'f32-mfma'), (2, 'planes 128x128'), (3, 'planes 256x128 db')]:
        f = lambda ldc=N: _lib.check(lib.mi_debug_gemm(kind, A.data_ptr(), K, W.data_ptr(), K, Cm.data_ptr(), ldc, M, N, K, None))
        f(); torch.cuda.synchronize()
        g = (lambda: f(-N)) if kind >= 2 else f   # kind 2 with negative ldc: reuse the planes, time the GEMM only
        g(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        err = (Cm.double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"{M}x{N}x{K} {name:18s} {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TF-eq  err {err:.2e}")
