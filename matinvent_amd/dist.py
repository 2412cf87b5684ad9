"""Data-parallel helpers (one process per GPU; torch.distributed over RCCL on the device,
gloo in the CPU tests).  The path shards by crystal: independent units, so the sampler needs no
data-path collective and the fine-tune step needs exactly one flat all-reduce per optimizer step."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_range(n: int, rank: int, world: int):
    """Contiguous block [lo, hi) of n units for `rank`: the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allreduce_flat_(buf: torch.Tensor):
    """In-place SUM all-reduce of one flat buffer (the whole gradient: 4P bytes, one message)."""
    if is_dist() and dist.get_world_size() > 1:
        if buf.is_cuda and dist.get_backend() == "gloo":  # CPU-backend test runs: stage through the host
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def all_gather_objects(obj):
    if not is_dist() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def broadcast_object(obj, src=0):
    if not is_dist() or dist.get_world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]
