"""Data-parallel helpers (one process per GPU; torch.distributed over RCCL on the device,
gloo in the CPU tests).  The path shards by crystal: independent units, so the sampler needs no
data-path collective and the fine-tune step needs exactly one flat all-reduce per optimizer step."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def collectives_on():
    """True when the collectives of this module really run: a process group exists and it has more than one rank -- or it has ONE rank and
    MI_DIST_FORCE_COLLECTIVES=1 asks for them anyway.  A world-size-1 RCCL group executes every line of the multi-GPU path with its real
    arguments (device buffers, device ids) on a one-GPU box: tests/test_gpu_multirank.py::test_world_size_one_rccl_group_runs_every_collective."""
    import os
    return is_dist() and (dist.get_world_size() > 1 or os.environ.get("MI_DIST_FORCE_COLLECTIVES", "0") not in ("", "0"))


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_range(n: int, rank: int, world: int):
    """Contiguous block [lo, hi) of n units for `rank`: the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _trace_lib():
    """The HIP library when roctx ranges are on (MI_ROCTX=1), else None: the CPU-only gloo tests never load it."""
    import os
    if os.environ.get("MI_ROCTX", "0") in ("", "0"):
        return None
    from . import _lib
    return _lib.load()


def _wire_of(buf: torch.Tensor, backend: str):
    """The tensor that goes on the wire for `buf`: the device buffer itself under RCCL ("nccl": xGMI moves device memory); a pinned
    host copy under gloo (the CPU tests, and the shared-GPU multi-rank tests: gloo cannot read device memory)."""
    if buf.is_cuda and backend != "nccl":
        wire = torch.empty(buf.shape, dtype=buf.dtype, device="cpu", pin_memory=True)
        wire.copy_(buf, non_blocking=True)
        torch.cuda.current_stream(buf.device).synchronize()
        return wire
    return buf


def _unwire(buf: torch.Tensor, wire: torch.Tensor):
    if wire is not buf:
        buf.copy_(wire, non_blocking=True)
    return buf


def allreduce_flat_(buf: torch.Tensor):
    """In-place SUM all-reduce of one flat buffer (the whole gradient: 4P bytes, one message; pipeline/mat_invent.py:166,177 is where the
    reference steps its optimizer -- the all-reduce sits right in front).  ONE code path for every backend: stage (a no-op under RCCL),
    reduce, unstage; the only line a gloo run does not execute with the arguments of an RCCL run is the collective itself."""
    if not collectives_on():
        return buf
    assert buf.is_contiguous() and buf.dtype == torch.float32, "the flat gradient / accumulator buffer is one contiguous fp32 vector"
    lib = _trace_lib()
    if lib is not None:
        lib.mi_trace_push(b"mi_grad_allreduce")
    try:
        wire = _wire_of(buf, dist.get_backend())
        dist.all_reduce(wire, op=dist.ReduceOp.SUM)
        return _unwire(buf, wire)
    finally:
        if lib is not None:
            lib.mi_trace_pop()


def all_gather_objects(obj):
    if not collectives_on():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def broadcast_object(obj, src=0):
    if not collectives_on():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]
