"""CPU-only dry runs of bench.py's multi-GPU launch path (no 8-GPU node has been available to any round; the driver runs
`--gpus 1,2,4,8` at round end): what the rank launcher starts, what every rank does with its environment, and that the all-reduce
helper takes one code path for every backend."""
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture()
def bench(monkeypatch):
    import importlib
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    return importlib.import_module("bench")


def test_self_launch_starts_one_rank_per_gpu_on_the_loopback_address(bench, monkeypatch):
    import subprocess
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    assert bench._self_launch(8) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]   # the ranks get the caller's flags unchanged
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"               # (dmabuf IPC: RCCL fails with the legacy mode on this host driver)


def _rank_env(monkeypatch, world, rank):
    for k, v in dict(WORLD_SIZE=world, RANK=rank, LOCAL_RANK=rank, MASTER_ADDR="127.0.0.1", MASTER_PORT=29511).items():
        monkeypatch.setenv(k, str(v))
    monkeypatch.delenv("MI_BENCH_SHARE_GPU", raising=False)


def test_every_rank_binds_its_own_device_and_opens_rccl(bench, monkeypatch):
    import torch.distributed as dist
    calls = {"set_device": [], "init": []}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls["set_device"].append(d))
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: calls["init"].append((backend, kw)))
    for rank in (0, 5, 7):
        calls["set_device"].clear()
        calls["init"].clear()
        _rank_env(monkeypatch, 8, rank)
        world, r, local_rank, share, d = bench.dist_setup(8)
        assert (world, r, local_rank, share) == (8, rank, rank, False) and d is dist
        assert calls["init"] == [("nccl", {"device_id": torch.device("cuda", rank)})]   # backend "nccl" IS RCCL on ROCm
        assert calls["set_device"][0] == rank and calls["set_device"][-1] == torch.device("cuda", rank)
        assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_fewer_gpus_than_ranks_is_refused_with_a_message(bench, monkeypatch):
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    _rank_env(monkeypatch, 8, 6)
    with pytest.raises(SystemExit, match="8 ranks .* 4 GPU"):
        bench.dist_setup(8)
    _rank_env(monkeypatch, 4, 1)
    with pytest.raises(SystemExit, match="--gpus 8 but WORLD_SIZE=4"):
        bench.dist_setup(8)


def test_the_all_reduce_helper_stages_only_what_the_backend_cannot_read(monkeypatch):
    """allreduce_flat_ = stage -> dist.all_reduce -> unstage for every backend.  Under RCCL ("nccl") the wire tensor IS the device buffer
    (no staging); under gloo a device buffer goes through pinned host memory.  With the collective stubbed, both branches run here."""
    from matinvent_amd import dist as D
    seen = []
    fake = types.SimpleNamespace(is_available=lambda: True, is_initialized=lambda: True, get_world_size=lambda: 2, get_rank=lambda: 0,
                                 ReduceOp=types.SimpleNamespace(SUM="sum"), get_backend=lambda: "nccl",
                                 all_reduce=lambda t, op=None: (seen.append((t, op)), t.mul_(2.0))[0])
    monkeypatch.setattr(D, "dist", fake)
    buf = torch.arange(8, dtype=torch.float32)
    out = D.allreduce_flat_(buf)
    assert out is buf and seen[0][0] is buf and seen[0][1] == "sum" and torch.equal(buf, 2.0 * torch.arange(8, dtype=torch.float32))
    assert D._wire_of(buf, "nccl") is buf and D._wire_of(buf, "gloo") is buf      # host buffers are never staged
    with pytest.raises(AssertionError):
        D.allreduce_flat_(torch.arange(8, dtype=torch.float64))
    fake.get_world_size = lambda: 1
    seen.clear()
    D.allreduce_flat_(buf)
    assert not seen   # a single rank never enters the collective


def test_strong_scaling_shards_one_global_batch_of_256(bench):
    """`--scaling strong`: the global batch of 256 crystals is cut over the ranks (north_star's "batch 256 ... at 8 GPUs"); weak (default) keeps 256 per rank;
    `--batch` measures a per-rank shape on one GPU; `--streams 0` takes the sampler's automatic chain count for that shape."""
    ns = lambda **kw: types.SimpleNamespace(**dict(dict(gpus=1, scaling="weak", batch=0, streams=4), **kw))
    a = ns(gpus=8)
    assert bench.resolve_batch(a) == 256 and "256 crystals per GPU" in a.batch_label
    a = ns(gpus=8, scaling="strong")
    assert bench.resolve_batch(a) == 32 and bench.B == 32 and "sharded over 8 rank(s): 32 crystals per GPU" in a.batch_label
    a = ns(gpus=2, scaling="strong", streams=0)
    assert bench.resolve_batch(a) == 128 and a.streams == 3
    a = ns(batch=64, streams=0)
    assert bench.resolve_batch(a) == 64 and a.streams == 2 and "per-rank shape" in a.batch_label
    with pytest.raises(SystemExit, match="does not divide"):
        bench.resolve_batch(ns(gpus=3, scaling="strong"))
    bench.resolve_batch(ns())     # (back to the default for whoever imports the module next)
    assert bench.B == 256
