"""Steady-state rate of the 128 x 128 plane product in its three operand-staging forms (register-staged, one set; LDS-DMA; large-M
256 x 256 LDS-DMA) on the shapes of the pinned path and a cube: time per launch and issued fp16 PF/s (3 terms)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matinvent_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
for (M, N, K) in [(25600, 512, 512), (102400, 512, 512), (102400, 512, 768), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5
    Cm = torch.empty(M, N, device='cuda')
    for name, dma, big in (("register-staged 128x128", 0, 0), ("LDS-DMA 128x128", 2, 0), ("LDS-DMA 256x256", 0, 1)):
        _lib.check(lib.mi_debug_set_planes_latency(0)); _lib.check(lib.mi_debug_set_planes_dma(dma)); _lib.check(lib.mi_debug_set_planes_big(big, 1))
        f = lambda ldc=N: _lib.check(lib.mi_debug_gemm(2, A.data_ptr(), K, W.data_ptr(), K, Cm.data_ptr(), ldc, M, N, K, None))
        f(); torch.cuda.synchronize()
        f(-N); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f(-N)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{M}x{N}x{K} {name:24s} {dt*1e6:8.1f} us  {3*2*M*N*K/dt/1e15:5.2f} PF/s issued", flush=True)
