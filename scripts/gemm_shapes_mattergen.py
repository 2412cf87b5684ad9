import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from matinvent_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
for (M, N, K) in [(256000, 512, 512), (256000, 64, 512), (256000, 512, 64), (256000, 64, 1024), (256000, 512, 128), (256000, 512, 16)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5
    Cm = torch.empty(M, N, device='cuda')
    for kind, name in [(1, 'fp32-operand split'), (2, 'planes 128x128')]:
        f = lambda ldc=N: _lib.check(lib.mi_debug_gemm(kind, A.data_ptr(), K, W.data_ptr(), K, Cm.data_ptr(), ldc, M, N, K, None))
        f(); torch.cuda.synchronize()
        g = (lambda: f(-N)) if kind >= 2 else f
        g(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{M}x{N}x{K} {name:18s} {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TF-eq")
