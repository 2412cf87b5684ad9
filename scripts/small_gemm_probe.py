"""Where the ~36 us of a one-round plane product go: the same small product launched back to back (warm instruction cache) and
alternating with other kernels (as in the network), latency form on / off.  Run under rocprofv3 --kernel-trace for per-kernel times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matinvent_amd import _lib
lib = _lib.load()
torch.manual_seed(0)
for (M, N, K) in [(265, 512, 768), (265, 512, 512), (31, 1536, 512), (7500, 512, 512)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5
    Cm = torch.empty(M, N, device='cuda')
    A2 = torch.randn(64, 512, device='cuda'); W2 = torch.randn(512, 512, device='cuda'); C2 = torch.empty(64, 512, device='cuda')
    for lat in (0, 256):
        _lib.check(lib.mi_debug_set_planes_latency(lat))
        f = lambda ldc=N: _lib.check(lib.mi_debug_gemm(2, A.data_ptr(), K, W.data_ptr(), K, Cm.data_ptr(), ldc, M, N, K, None))
        other = lambda: (_lib.check(lib.mi_debug_gemm(1, A2.data_ptr(), 512, W2.data_ptr(), 512, C2.data_ptr(), 512, 64, 512, 512, None)), torch.nn.functional.layer_norm(C2, (512,)))
        f(); other(); torch.cuda.synchronize()
        res = []
        for mode in ("back-to-back", "alternating"):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for _ in range(50):
                if mode == "alternating":
                    other()
                ev0.record(); f(-N); ev1.record(); torch.cuda.synchronize()
                tot += ev0.elapsed_time(ev1)
            res.append(tot / 50 * 1e3)
        print(f"{M}x{N}x{K} latency-form={lat:3d}: back-to-back {res[0]:6.1f} us, after other kernels {res[1]:6.1f} us (event-bracketed single launches)")
