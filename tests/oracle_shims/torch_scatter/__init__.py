"""Pure-torch stand-in for the three torch_scatter symbols the reference uses.

scatter(src, index, dim, reduce, dim_size): 'sum' / 'mean' (count clamped >= 1,
empty segment -> 0), as documented by torch_scatter.
segment_coo / segment_csr: sum reductions over sorted index / CSR pointer.
"""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim in (0, -src.dim()), "shim only supports dim=0"
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = (dim_size,) + tuple(src.shape[1:])
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    res.index_add_(0, index, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp(min=1)
        return res / cnt.view((-1,) + (1,) * (src.dim() - 1))
    raise NotImplementedError(reduce)


def segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    assert reduce in ("sum", "add")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    res = torch.zeros((int(dim_size),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res.index_add_(0, index, src)
    return res


def segment_csr(src, indptr, out=None, reduce="sum"):
    assert reduce in ("sum", "add")
    n = indptr.numel() - 1
    lens = indptr[1:] - indptr[:-1]
    idx = torch.repeat_interleave(torch.arange(n, device=src.device), lens)
    start = int(indptr[0])
    seg = src[start:start + idx.numel()]
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res.index_add_(0, idx, seg)
    return res
