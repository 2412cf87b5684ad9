"""The three GEMM kernels behind every linear of the path, against fp64: the bf16 three-plane split
(on-the-fly and pre-split/tile-blocked) must be at least as accurate as the f32-input MFMA."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(300, 100, 100), (5120, 512, 1024), (20000, 512, 768), (129, 512, 48)])
def test_gemm_kernels_vs_fp64(M, N, K):
    from matinvent_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    ref = A.double() @ W.double().t()
    scale = ref.abs().max().item()
    errs = {}
    for kind in (0, 1, 2):
        out = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.mi_debug_gemm(kind, C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K, C.c_void_p(out.data_ptr()), N, M, N, K, None))
        torch.cuda.synchronize()
        errs[kind] = (out.double() - ref).abs().max().item() / scale
        assert errs[kind] < 3e-6, (kind, errs)
    # fp32-class: the split paths are not worse than the f32 MFMA by more than round-off noise
    assert errs[1] <= 1.5 * errs[0] + 1e-7 and errs[2] <= 1.5 * errs[0] + 1e-7, errs


@pytest.mark.parametrize("M,N,K", [(33333, 512, 768), (66000, 512, 96), (33000, 512, 48), (140000, 64, 64)])
def test_double_buffered_plane_gemm_vs_fp64(M, N, K):
    """The 256x128 double-buffered kernel (kind 3) on shapes that engage it: ragged M (not a multiple of 256 or 128, odd number
    of 128-row tiles), odd and padded k-tile counts, a single narrow column tile."""
    from matinvent_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    ref = A.double() @ W.double().t()
    scale = ref.abs().max().item()
    errs = {}
    for kind in (2, 3):
        out = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.mi_debug_gemm(kind, C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K, C.c_void_p(out.data_ptr()), N, M, N, K, None))
        torch.cuda.synchronize()
        errs[kind] = (out.double() - ref).abs().max().item() / scale
        assert errs[kind] < 3e-6, (kind, errs)


@pytest.mark.parametrize("M,N,K", [(1280, 1536, 512), (5120, 512, 512), (193, 512, 96), (64, 100, 100)])
def test_plane_gemm_64_row_tiles_bit_identical(M, N, K):
    """Few-tile products (the node-level ones) run on 64-row tiles; per output element the k order and the order of the six
    product terms are those of the 128-row kernel, so the two must agree bit for bit (ragged M: odd number of 64-row tiles)."""
    from matinvent_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    outs = []
    try:
        for small in (0, 1 << 30):
            _lib.check(lib.mi_debug_set_planes_small_tiles(small))
            out = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(lib.mi_debug_gemm(2, C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K, C.c_void_p(out.data_ptr()), N, M, N, K, None))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        _lib.check(lib.mi_debug_set_planes_small_tiles(0))
    assert torch.equal(outs[0], outs[1])
    ref = A.double() @ W.double().t()
    assert (outs[1].double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6


@pytest.mark.parametrize("M,N,K", [(512, 512, 20000), (512, 384, 48640), (200, 130, 9000), (512, 1024, 5120), (100, 3, 700)])
def test_weight_gradient_gemm_vs_fp64(M, N, K):
    """C += A^T W over K rows (the dW products of the backward pass): the bf16 three-plane-split kernel (long row lists, default),
    the 128x128-tile f32-MFMA kernel and the 64x64 one must all accumulate into C and agree with fp64; ragged tile edges and
    unaligned leading dimensions included."""
    from matinvent_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g).cuda()
    W = torch.randn(K, N, generator=g).cuda()
    C0 = torch.randn(M, N, generator=g).cuda()
    ref = C0.double() + A.double().t() @ W.double()
    scale = ref.abs().max().item()
    try:
        for on in (0, 1, 3):
            _lib.check(lib.mi_debug_set_tn128(on))
            out = C0.clone()
            _lib.check(lib.mi_debug_gemm(4, C.c_void_p(A.data_ptr()), M, C.c_void_p(W.data_ptr()), N, C.c_void_p(out.data_ptr()), N, M, N, K, None))
            torch.cuda.synchronize()
            assert (out.double() - ref).abs().max().item() / scale < 3e-6, on
    finally:
        _lib.check(lib.mi_debug_set_tn128(3))


@pytest.mark.parametrize("M,N,K", [(512, 512, 25600), (256, 384, 12163), (512, 128, 4099)])
def test_weight_gradient_from_plane_sets_vs_fp64(M, N, K):
    """C[M, N] += A^T W over a long row list with both operands given as fp16 plane sets (csrc/backward.hip gemm_tn_planes_kernel: LDS-DMA slabs,
    the k-strided MFMA operands through ds_read_b64_tr_b16): against the fp64 product, at the plane format's accuracy; row counts that are not
    multiples of the 32-row slab (the tail reads the plane sets' zero row padding) and both tile counts per operand."""
    from matinvent_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g).cuda()
    W = torch.randn(K, N, generator=g).cuda()
    C0 = torch.randn(M, N, generator=g).cuda()
    ref = C0.double() + A.double().t() @ W.double()
    scale = ref.abs().max().item()
    out = C0.clone()
    _lib.check(lib.mi_debug_gemm(5, C.c_void_p(A.data_ptr()), M, C.c_void_p(W.data_ptr()), N, C.c_void_p(out.data_ptr()), N, M, N, K, None))
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() / scale < 3e-6   # (the fp32-row forms' bound in the test above)


@pytest.mark.parametrize("M,N,K", [(3000, 512, 512), (300, 256, 96), (1031, 768, 64), (256, 512, 16), (70000, 512, 512)])
def test_plane_gemm_lds_dma_256_tiles_bit_identical(M, N, K):
    """Large plain products run on 256 x 256 tiles whose operands arrive by LDS-DMA into two LDS stages; the k order and the order
    of the product terms per output are the 128-row kernel's, so the two agree bit for bit (ragged M: a workgroup whose second
    128-row half lies outside the plane set reads zeros through its buffer descriptor; K = 16: a single k-tile)."""
    from matinvent_amd import _lib
    lib = _lib.load()
    if lib.mi_plane_format() != 2:
        pytest.skip("the LDS-DMA kernel exists in the fp16 two-plane build")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    outs = []
    try:
        _lib.check(lib.mi_debug_set_planes_rt(0, 0))   # (the register-tile form would take these shapes first)
        for big in (0, 1):
            _lib.check(lib.mi_debug_set_planes_big(big, 1))
            out = torch.full((M, N), float("nan"), device="cuda")
            for _ in range(3 if big else 1):   # (repeated: a DMA / barrier ordering slip would show as run-to-run differences)
                _lib.check(lib.mi_debug_gemm(2, C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K, C.c_void_p(out.data_ptr()), N, M, N, K, None))
                torch.cuda.synchronize()
                outs.append(out.clone())
    finally:
        _lib.check(lib.mi_debug_set_planes_big(1, 65536))
        _lib.check(lib.mi_debug_set_planes_rt(2, 0))
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    ref = A.double() @ W.double().t()
    assert (outs[1].double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6


@pytest.mark.parametrize("M,N,K", [(3000, 512, 512), (129, 512, 512), (1031, 768, 128), (300, 256, 192), (70000, 512, 512)])
def test_plane_gemm_register_tile_form_bit_identical(M, N, K):
    """gemm_rt (csrc/edge_stage.hip): 128 rows x 256 columns per four-wave workgroup, W in MFMA fragment order from L2 into a register
    ring, A through four LDS stages by LDS-DMA, the plane GEMM's own row-major epilogue.  Same product terms in the same k order per
    output as the 128 x 128 kernel: bit for bit, ragged M and the shortest K (four k-tiles: the prologue and the drain meet) included."""
    from matinvent_amd import _lib
    lib = _lib.load()
    if lib.mi_plane_format() != 2:
        pytest.skip("the register-tile kernel exists in the fp16 two-plane build")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    outs = []
    try:
        _lib.check(lib.mi_debug_set_planes_big(0, 1))
        for rt in (0, 2):
            _lib.check(lib.mi_debug_set_planes_rt(rt, 1))
            out = torch.full((M, N), float("nan"), device="cuda")
            for _ in range(3 if rt else 1):
                _lib.check(lib.mi_debug_gemm(2, C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K, C.c_void_p(out.data_ptr()), N, M, N, K, None))
                torch.cuda.synchronize()
                outs.append(out.clone())
    finally:
        _lib.check(lib.mi_debug_set_planes_rt(2, 16384))
        _lib.check(lib.mi_debug_set_planes_big(1, 65536))
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    ref = A.double() @ W.double().t()
    assert (outs[1].double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6


@pytest.mark.parametrize("M,N,K", [(265, 512, 768), (31, 1536, 512), (300, 256, 96), (128, 128, 16), (1000, 384, 64), (700, 512, 128), (513, 512, 160),
                                   (3000, 512, 512), (70000, 512, 512)])
def test_plane_gemm_latency_and_lds_dma_forms_bit_identical(M, N, K):
    """Launches of at most one workgroup per CU run a latency-tolerant form of the 128 x 128 kernel: by default the LDS-DMA form
    (operand tiles by `buffer_load ... lds` into two stages, fragments software-pipelined over two register sets, one barrier per
    k-tile), optionally the register form (three operand register sets, loads three k-tiles ahead); mode 2 runs the LDS-DMA form for
    every launch.  The k order and the order of the product terms per output are unchanged, so all agree bit for bit with the
    one-set loop -- k-tile counts of 1, 2, 3 and more, odd and even, ragged M and N, single- and multi-round launches."""
    from matinvent_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    outs = []
    try:
        _lib.check(lib.mi_debug_set_planes_big(0, 1))
        # (modes 3 / 4 -- recorded ablations: the LDS-DMA form for the LARGE launches only, one-round launches register-staged / on the
        #  four-waves-per-SIMD build)
        for dma, lat in ((0, 0), (0, 256), (1, 256), (2, 0), (3, 256), (4, 256)):
            if lib.mi_debug_set_planes_dma(dma) != 0:   # (mode 4 is an ablation instantiation that spills: only in a -DMI_ABLATION_KERNELS build)
                continue
            _lib.check(lib.mi_debug_set_planes_latency(lat))
            for _ in range(3 if dma else 1):   # (repeated: a DMA / barrier ordering slip would show as run-to-run differences)
                out = torch.full((M, N), float("nan"), device="cuda")
                _lib.check(lib.mi_debug_gemm(2, C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K, C.c_void_p(out.data_ptr()), N, M, N, K, None))
                torch.cuda.synchronize()
                outs.append(out)
    finally:
        _lib.check(lib.mi_debug_set_planes_latency(256))
        _lib.check(lib.mi_debug_set_planes_dma(1))
        _lib.check(lib.mi_debug_set_planes_big(1, 65536))
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    ref = A.double() @ W.double().t()
    assert (outs[1].double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6


@pytest.mark.parametrize("na", [[10, 7, 4, 10, 20, 1], [20] * 40], ids=["one-round-launches", "multi-round-launches"])
def test_latency_forms_in_the_network_bit_identical(na):
    """A short chain at the benchmark network (pair-mode Fourier GEMM with its folded self edges and scales, the second linear with
    the fused segmented sum, the node-level products) with the plane products on the one-set loop, the register latency form, the
    LDS-DMA form for one-round launches (default) and the LDS-DMA form everywhere: the final states must be equal bit for bit."""
    from matinvent_amd import _lib
    from tests.gpu_util import Box, make_module
    lib = _lib.load()
    # (this test is about the forms of the PLANE GEMM: the node-level chain and the second edge GEMM stay on it -- the one-launch node chain
    #  and the register-tile second GEMM of round 3 have their own comparison in tests/test_gpu_forward.py)
    lib.mi_debug_set_node_fused(0)
    lib.mi_debug_set_edge2_fused(0)
    torch.manual_seed(0)
    m = make_module(512, 6, 128, 1000)
    with torch.no_grad():
        v = m.decoder.views()
        for k in ("coord_out.weight", "lattice_out.weight", "type_out.weight", "type_out.bias"):
            v[k].mul_(1e-2)
    m.decoder.mark_dirty()
    finals = []
    try:
        for dma, lat, bigseg, hi in ((0, 0, 0, 0), (0, 256, 0, 0), (1, 256, 0, 0), (2, 0, 0, 0), (1, 256, 1, 0), (1, 256, 0, 1), (3, 256, 0, 0), (4, 256, 0, 0)):
            # (the instantiations that spill registers -- dma mode 4, the big-tile segmented-sum epilogue -- exist only in a
            #  -DMI_ABLATION_KERNELS build: their switches refuse otherwise, and the configuration is skipped)
            if lib.mi_debug_set_planes_dma(dma) != 0 or lib.mi_debug_set_planes_big_seg(bigseg) != 0:
                lib.mi_debug_set_planes_dma(1)
                continue
            _lib.check(lib.mi_debug_set_planes_latency(lat))
            _lib.check(lib.mi_debug_set_node_priority(hi))        # (1: node-level kernels on the batch's high-priority helper stream, joined by events)
            final, _ = m.sample(Box(na), seed=5, step_lr=5e-6, t_start=1000, t_stop=997, streams=1)
            torch.cuda.synchronize()
            finals.append({k: v.clone() for k, v in final.items() if torch.is_tensor(v)})
    finally:
        _lib.check(lib.mi_debug_set_planes_latency(256))
        _lib.check(lib.mi_debug_set_planes_dma(1))
        _lib.check(lib.mi_debug_set_planes_big_seg(0))
        _lib.check(lib.mi_debug_set_node_priority(0))
        lib.mi_debug_set_node_fused(1)
        lib.mi_debug_set_edge2_fused(1)
    for f in finals[1:]:
        for k in ("frac_coords", "atom_types", "lattices"):
            assert torch.isfinite(finals[0][k]).all()
            assert torch.equal(finals[0][k], f[k]), k
