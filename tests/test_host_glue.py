"""CPU-only checks of the host glue against reference-generated fixtures (g3, g9) and a
world_size-2 gloo run of the data-parallel fine-tune arithmetic."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_atom_count_prior_and_step_lr(golden):
    from matinvent_amd.sampling import ATOM_DIST, DEFAULT_STEP_LR, SampleDataset
    g = golden("g9_host_glue")
    np.testing.assert_array_equal(np.array(ATOM_DIST["mp_20"]), g["atom_dist_mp20"])
    assert DEFAULT_STEP_LR["gen"]["mp_20"] == float(g["step_lr_gen_mp20"])
    np.random.seed(0)
    assert SampleDataset(8).num_atoms.tolist() == g["num_atoms_seed0_8"].tolist()
    np.random.seed(0)
    assert SampleDataset(64).num_atoms.tolist() == g["num_atoms_seed0_64"].tolist()


def test_lattices_to_params_shape(golden):
    from matinvent_amd.data import lattices_to_params_shape
    g = golden("g3_lattice")
    l, a = lattices_to_params_shape(torch.from_numpy(g["rnd"]))
    np.testing.assert_array_equal(l.numpy(), g["rnd_lengths"])
    np.testing.assert_array_equal(a.numpy(), g["rnd_angles"])


def test_collate_and_dataset():
    from matinvent_amd.data import CrystalBatchData, CrystalData, CrystalDataset, CrystalLoader
    items = [CrystalData(torch.rand(n, 3), torch.randint(1, 95, (n,)), torch.rand(1, 3) + 4, torch.full((1, 3), 90.0)) for n in (3, 1, 5)]
    ds = CrystalDataset(items, rewards=np.array([0.1, 0.5, 0.9]))
    b = CrystalBatchData([ds[i] for i in range(3)])
    assert b.num_graphs == 3 and b.num_nodes == 9 and b.batch.tolist() == [0, 0, 0, 1, 2, 2, 2, 2, 2]
    assert b.reward.tolist() == [np.float32(0.1), np.float32(0.5), np.float32(0.9)]
    assert b.lengths.shape == (3, 3) and b.atom_types.dtype == torch.long
    torch.manual_seed(0)
    batches = list(CrystalLoader(ds, batch_size=2, shuffle=True))
    assert sum(x.num_graphs for x in batches) == 3 and len(batches) == 2


def test_shard_range_partitions():
    from matinvent_amd.dist import shard_range
    for n in (0, 1, 7, 18, 256, 2048):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_schedules_match_reference_fixture(golden):
    from matinvent_amd.schedules import BetaScheduler, SigmaScheduler, sampler_coefficients
    g = golden("g2_schedulers")
    for Tn in (20, 1000):
        b = BetaScheduler(Tn, "cosine")
        for k in ("betas", "alphas", "alphas_cumprod", "sigmas"):
            np.testing.assert_allclose(getattr(b, k).numpy(), g[f"T{Tn}_beta_{k}"], rtol=2e-6, atol=1e-7)
        torch.manual_seed(1234)
        s = SigmaScheduler(Tn, 0.005, 0.5)
        np.testing.assert_allclose(s.sigmas.numpy(), g[f"T{Tn}_sigma_sigmas"], rtol=1e-7)
        np.testing.assert_allclose(s.sigmas_norm.numpy(), g[f"T{Tn}_sigma_sigmas_norm"], rtol=1e-5)
    for mode in ("linear", "quadratic", "sigmoid"):
        np.testing.assert_allclose(BetaScheduler(50, mode).betas.numpy(), g[f"T50_{mode}_betas"], rtol=1e-6, atol=1e-9)
    coef = sampler_coefficients(BetaScheduler(20, "cosine"), SigmaScheduler(20, 0.005, 0.5, sigmas_norm=torch.ones(21)), 5e-6)
    assert coef.shape == (21, 16) and torch.isfinite(coef[2:]).all()


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from oracle import diffcsp_oracle as O
from matinvent_amd.dist import shard_range, allreduce_flat_
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.set_num_threads(2)
hp = O.CSPNetHParams(hidden_dim=64, num_layers=1, num_freqs=8)
P = O.init_params(hp, seed=0)
sch = O.Schedules.make(50, sigmas_norm=torch.ones(51))
g = torch.Generator().manual_seed(1)
na = torch.tensor([3, 5, 2, 4, 6])
B, N = len(na), int(na.sum())
full = dict(num_atoms=na, lengths=4 + 6 * torch.rand(B, 3, generator=g), angles=70 + 40 * torch.rand(B, 3, generator=g),
            frac_coords=torch.rand(N, 3, generator=g), atom_types=torch.randint(1, 95, (N,), generator=g))
rewards = torch.rand(B, generator=g)
noise = dict(rand_l=torch.randn(B, 3, 3, generator=g), rand_x=torch.randn(N, 3, generator=g), rand_t=torch.randn(N, 100, generator=g))
accum, sigma = 2, 0.025
names = list(P)

def grad_of(batch, rw, nz, denom):
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    noised = O.add_noise(hp, sch, batch, 7, nz)
    loss_b, pred = O.calc_sample_loss(Pg, hp, O.Costs(), noised)
    with torch.no_grad():
        _, pp = O.calc_sample_loss({k: v * 1.01 for k, v in P.items()}, hp, O.Costs(), noised)
    kl = O.calc_kl_reg(pred, pp, noised[2], len(rw))
    loss = (rw * loss_b + kl * (1.1 - rw) * sigma).sum() / denom
    gs = torch.autograd.grad(loss, [Pg[k] for k in names])
    return torch.cat([x.reshape(-1) for x in gs])

lo, hi = shard_range(B, rank, world)
off = [0] + torch.cumsum(na, 0).tolist()
sl = slice(off[lo], off[hi])
shard = dict(num_atoms=na[lo:hi], lengths=full["lengths"][lo:hi], angles=full["angles"][lo:hi],
             frac_coords=full["frac_coords"][sl], atom_types=full["atom_types"][sl])
nz = dict(rand_l=noise["rand_l"][lo:hi], rand_x=noise["rand_x"][sl], rand_t=noise["rand_t"][sl])
# local sum / (B_global * accum), then ONE flat all-reduce  == reference .mean() / accum on the whole batch
flat = grad_of(shard, rewards[lo:hi], nz, B * accum)
allreduce_flat_(flat)
ref = grad_of(full, rewards, noise, B * accum)
err = float((flat - ref).abs().max() / ref.abs().max())
assert err < 2e-5, err
if rank == 0:
    print("DP_OK", err)
dist.destroy_process_group()
'''


def test_data_parallel_gradient_equals_full_batch_gloo(tmp_path):
    """world_size 2 over gloo: sharding by crystal with sum/(B_global*accum) scaling and one flat
    all-reduce reproduces the full-batch gradient of the reference's `.mean()/accum` loss."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DP_OK" in out.stdout


def test_replay_buffer_follows_the_reference_semantics():
    """memory/replay_buffer.py:33-105: merge, dedupe on the reduced formula keeping the best reward, top-K, THEN the strict
    reward cutoff; sample size default 8; purge by formula."""
    from types import SimpleNamespace as NS
    from matinvent_amd.memory import ReplayBuffer
    mk = lambda types: NS(atom_types=np.array(types))
    rb = ReplayBuffer(buffer_size=3, reward_cutoff=0.2)
    assert rb.sample_size == 8
    fe2o3, fe4o6, nacl, sio2, h2 = mk([26, 26, 8, 8, 8]), mk([26] * 4 + [8] * 6), mk([11, 17]), mk([14, 8, 8]), mk([1, 1])
    rb.extend([fe2o3, nacl, h2], None, [0.5, 0.2, 0.9])
    # H2 0.9, Fe2O3 0.5 stay; NaCl sits exactly on the cutoff and is dropped (strict >)
    assert [round(r[0], 3) for r in rb.rows] == [0.9, 0.5]
    rb.extend([fe4o6, sio2], None, [0.7, 0.6])
    # Fe4O6 has the same reduced formula as Fe2O3 and the higher reward: it replaces it; top-3 = H2, Fe4O6, SiO2
    assert [round(r[0], 3) for r in rb.rows] == [0.9, 0.7, 0.6] and rb.rows[1][2] is fe4o6
    data, rewards = rb.sample()
    assert len(data) == 3 and sorted(np.round(rewards, 3).tolist()) == [0.6, 0.7, 0.9]
    rb.memory_purge([NS(species=[8, 14, 8])])
    assert [round(r[0], 3) for r in rb.rows] == [0.9, 0.7]


def test_checkpoint_choice_is_numeric_on_the_epoch(tmp_path):
    """models/suite/diffcsp.py:83-90: without a `last` file the numerically largest epoch wins (epoch=10 over epoch=9)."""
    names = ["epoch=9-step=1.ckpt", "epoch=10-step=2.ckpt"]
    epochs = [int(n.split("-")[0].split("=")[1]) for n in sorted(names)]
    assert sorted(names)[int(np.argsort(epochs)[-1])] == "epoch=10-step=2.ckpt"


def test_reward_scaling_modes_follow_the_reference(tmp_path):
    """rewards/reward.py:8-115: ascending / descending / float-target scaling, mean / min / weight reduction, NaN -> failed."""
    from types import SimpleNamespace as NS
    from matinvent_amd.rewards import PyMatGen, Reward, linear_scaling
    assert np.allclose(linear_scaling(np.array([-1.0, 3.0, 9.0])), [0.0, 0.5, 1.0])
    calc = lambda vals: NS(calc=lambda samples, label: np.array(vals, dtype=float))
    cfg = [dict(name="a", calculator=calc([750.0, 2000.0, 3250.0, np.nan]), target="descending", minv=750, maxv=3250, weight=0.25),
           dict(name="b", calculator=calc([0.0, 1.0, 2.5, 1.0]), target=2.5, minv=0.0, maxv=2.0, weight=0.75),
           dict(name="c", calculator=calc([2.0, 6.0, 12.0, 3.0]), target="ascending", minv=2.0, maxv=10.0, weight=1.0)]
    strucs = ([None] * 4, None)
    r, props, failed = Reward(str(tmp_path / "r"), cfg, 0.8, reduce="mean").scoring(strucs)
    # a: hhi-style descending -> [1, .5, 0]; b: |v - 2.5| = [2.5, 1.5, 0, 1.5] -> scale(-d, -2, 0) = [0, .25, 1, .25]; c: [0, .5, 1, .125]
    assert np.allclose(r, [(1 + 0 + 0) / 3, (0.5 + 0.25 + 0.5) / 3, (0 + 1 + 1) / 3, 0.0]) and failed.tolist() == [False, False, False, True]
    assert props["a"][3] == 0.0
    r, _, _ = Reward(str(tmp_path / "r"), cfg, 0.8, reduce="min").scoring(strucs)
    assert np.allclose(r, [0.0, 0.25, 0.0, 0.0])
    r, _, _ = Reward(str(tmp_path / "r"), cfg, 0.8, reduce="weight").scoring(strucs)
    assert np.allclose(r[:3], [0.25 * 1 + 0 + 0, 0.25 * 0.5 + 0.75 * 0.25 + 0.5, 0 + 0.75 + 1.0])
    # density calculator on a record: cubic 4 A cell with one Fe and one O
    s = NS(species=[26, 8], lengths=[4.0, 4.0, 4.0], angles=[90.0, 90.0, 90.0])
    d = PyMatGen(task="density").calc(([s], None))
    assert abs(d[0] - (55.845 + 15.999) * 1.66053906660 / 64.0) < 1e-6
    assert np.isnan(PyMatGen(task="hhi").calc(([s], None))[0])  # no table offline -> failed sample, like the reference's except branch


_W1 = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from matinvent_amd import dist as D
dist.init_process_group("gloo", rank=0, world_size=1)
x = torch.arange(10, dtype=torch.float32)
calls = []
orig = dist.all_reduce
dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), orig(t, *a, **k))[1]
assert not D.collectives_on()                       # a one-rank group runs no collective by default
D.allreduce_flat_(x)
assert D.all_gather_objects("a") == ["a"] and D.broadcast_object(3) == 3 and calls == []
os.environ["MI_DIST_FORCE_COLLECTIVES"] = "1"       # ... unless asked to: how the RCCL lines are executed on a one-GPU box
assert D.collectives_on()
y = D.allreduce_flat_(x.clone())
assert torch.equal(y, x) and calls == [10]          # SUM over one rank = identity, and the collective really ran
assert D.all_gather_objects({"k": 1}) == [{"k": 1}] and D.broadcast_object([1, 2]) == [1, 2]
dist.destroy_process_group()
print("ok")
'''


def test_forced_collectives_on_a_one_rank_group(tmp_path):
    """matinvent_amd.dist.collectives_on: a world-size-1 process group skips its collectives unless MI_DIST_FORCE_COLLECTIVES=1 -- the switch that lets ONE GPU execute
    every line of the RCCL path (tests/test_gpu_multirank.py::test_world_size_one_rccl_group_runs_every_collective, bench.py --force-dist).  Here on gloo / CPU."""
    import os
    import subprocess
    import sys
    script = tmp_path / "w1.py"
    script.write_text(_W1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("MI_DIST_FORCE_COLLECTIVES", None)
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
