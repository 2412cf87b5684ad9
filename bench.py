#!/usr/bin/env python
"""Headline benchmark: crystal structures / second of the 1000-step reverse-diffusion sampler.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "Config 2"): DiffCSP-architecture score
network H=512, L=6, F=128, fully connected edges, B=256 crystals x 20 atoms per GPU, T=1000,
two network evaluations per denoising step, synthetic random-init weights (seed 0, output
heads x1e-2), noise from the library's counter-based Philox stream (seed 1234), committed
sigmas_norm table.  The MatterGen (GemNet) arithmetic is un-vendored and parity-unpinned, so
the pinned DiffCSP network stands in at the same B / n / T.

A "step" is ONE denoising step of the chain over the whole batch (corrector eval + update +
predictor eval + update + log-probs).  Per-step cost does not depend on t, so
    value = n_gpus * B * K / (T * elapsed)          [complete 1000-step structures per second]
The default K = 1000 is one complete chain.  For N > 1 every rank samples its own B crystals
(independent units, no data-path collective): weak scaling.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
       python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, L, F, TD, T = 512, 6, 128, 256, 1000
B, NATOM = 256, 20
SEED_W, SEED_NOISE, HEAD_SCALE, STEP_LR = 0, 1234, 1e-2, 5e-6
SIGMAS_NORM = os.path.join(ROOT, "matinvent_amd", "data", "sigmas_norm_T1000_b0.005_e0.5_seed1234.npy")

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (the sparse figure is 2x)
SUSTAINED_F16_MFMA_TFLOPS = 1620.0  # what a chip-filling loop of v_mfma_f32_32x32x16_f16 on operands with random bits holds on this part (1.69 GHz, ~1285 W:
                                    # scripts/mfma_power.hip, profiles/r5_mfma_power.log) -- reported beside `peak`, never instead of it
PEAK_HBM_TBPS = 8.0


def edge_flops_per_edge(pairs=True):
    """fp32 flops per edge per layer in the edge-message stage.
    executed:    Fourier block (K = 6F) + second linear (K = H) on MFMA -- what the kernels issue.  On the default path the
                 Fourier block runs once per unordered atom pair (the reversed edge sees -sin / +cos of the same arguments):
                 n(n-1)/2 operand rows per crystal instead of n*n, self edges are a constant;
    algorithmic: SURVEY.md section 8(d): 2*(2H+9+6F)*H + 2*H*H (the reference's concat-GEMM form)."""
    rows = (NATOM - 1) / (2.0 * NATOM) if pairs else 1.0     # Fourier-GEMM rows per directed edge
    executed = rows * 2 * (6 * F) * H + 2 * H * H
    algorithmic = 2 * (2 * H + 9 + 6 * F) * H + 2 * H * H
    return executed, algorithmic


def bytes_per_crystal_eval(n):
    """SURVEY.md section 8(d) Y_eval(n): stage-boundary minimum HBM bytes per crystal-evaluation."""
    P = 12346468
    return 4 * (n * (103 + 103) + 265 + n * H * (1 + 5 * L + 1)) + 4 * P / B


def dist_setup(gpus=None):
    """One process per GPU (torch.distributed over RCCL = backend "nccl"; rendezvous from the launcher's environment).  Returns
    (world, rank, local_rank, share, dist).  MI_BENCH_SHARE_GPU (tests only): the N ranks share device 0 and talk over gloo, which
    exercises the control flow on a one-GPU box.  Fails with a clear message -- not a hang inside the collective -- when the box has
    fewer GPUs than ranks."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = bool(os.environ.get("MI_BENCH_SHARE_GPU"))
    dist = None
    if gpus is not None and world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: launch with torch.distributed.run (or from a bare shell, which self-launches)")
    force = world == 1 and os.environ.get("MI_BENCH_FORCE_DIST", "0") not in ("", "0")
    if force:
        # --force-dist: a WORLD-SIZE-1 RCCL process group, so that every line of the multi-GPU path -- init_process_group("nccl", device_id=...),
        # barrier(device_ids=...), the all-reduce of the device-resident time, the flat-gradient all-reduce on the real 49.4 MB device buffer --
        # runs with its real arguments on a one-GPU box.  The figures are those of a single GPU; `config.comm_backend` says "nccl".
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
        os.environ.setdefault("MASTER_PORT", str(port))
        os.environ["MI_DIST_FORCE_COLLECTIVES"] = "1"
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            n_dev = torch.cuda.device_count()
            if n_dev < world or local_rank >= n_dev:
                raise SystemExit(f"bench.py: {world} ranks (one process per GPU over RCCL) but this node shows {n_dev} GPU(s): "
                                 f"rank {rank} has no device {local_rank}.  Run with --gpus <= {n_dev}.")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(torch.device("cuda", local_rank))
    return world, rank, local_rank, share, dist


def dist_barrier(ctx):
    world, rank, local_rank, share, dist = ctx
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier() if share else dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()


def dist_max_time(ctx, elapsed):
    world, rank, local_rank, share, dist = ctx
    if dist is not None:
        tt = torch.tensor([elapsed], device="cpu" if share else torch.device("cuda", local_rank), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def mg_flops_eval(hp, N, E):
    """F_eval_mg: dense-layer flops (fp32 multiply-adds x 2) of ONE evaluation of the MatterGen-shaped network over N atoms and E directed
    edges (oracle/mattergen_oracle.py op by op): per edge the interaction blocks' 512-wide layers (concat, residual stacks, the
    triplet down / up projections and the bilinear layer) and the output blocks' layers, per atom the atom-update stacks and heads."""
    Ed, A = hp["emb_edge"], hp["emb_atom"]
    per_edge_block = 2 * Ed * (Ed * (2 + 2 * hp["num_before_skip"] + 2 * hp["num_after_skip"] + 1 + 2 * hp["num_concat"]) + hp["emb_rbf"] * 2
                               + hp["emb_trip"] + 2 * hp["emb_bil"]) + 2 * hp["emb_cbf"] * hp["emb_trip"] * hp["emb_bil"]
    # (output blocks: the force path's Dense + num_atom residual layers and the lattice head's Dense; the energy path is not evaluated -- E_t feeds no output)
    per_edge_out = 2 * Ed * (Ed * (2 + 2 * hp["num_atom"]) + 2 * hp["emb_rbf"])
    per_node_block = 2 * A * (Ed + A * 2 * hp["num_atom"] + 2 * Ed)
    return E * (hp["num_blocks"] * (per_edge_block + per_edge_out) + per_edge_out + 2 * hp["num_radial"] * (Ed + 3 * hp["emb_rbf"] + hp["num_spherical"] * hp["emb_cbf"])) \
        + N * (hp["num_blocks"] * per_node_block + 2 * A * (A + 2 * Ed + 101))


def mg_bytes_per_crystal_eval(hp, n, e, nparams, batch):
    """Y_eval_mg(n, e): stage-boundary MINIMUM HBM bytes (fp32) of one crystal-evaluation of the MatterGen-shaped network with n atoms and
    e directed edges -- the rules of SURVEY 8(d): weights once per evaluation amortised over the batch, every array that has to be
    globally visible at a message-passing boundary round-trips once, nothing else is written.  Unlike CSPNet the EDGE state is such an
    array: a triplet interaction gathers the down-projected messages of neighbouring edges, so the edge embedding m [e, emb_edge] is
    written by the embedding block and read + written once per interaction block, and the down-projection [e, emb_trip] is written and
    gathered once per block; node features round-trip once per block (+ embedding, + final read); the graph (src, dst, image code) is
    written once and read once; the two per-edge scalar heads of every output block are written and read once.  DESIGN 11."""
    nb, Ed, A, Tr = hp["num_blocks"], hp["emb_edge"], hp["emb_atom"], hp["emb_trip"]
    floats = n * (4 + 3 + 101) + 19 + 6 * e + e * Ed * (1 + 2 * nb) + 2 * nb * e * Tr + (2 * nb + 2) * n * A + 4 * (nb + 1) * e
    return 4 * floats + 4 * nparams / batch


def build_module(device):
    from matinvent_amd.diffcsp import DiffCSPModule
    torch.manual_seed(SEED_W)
    m = DiffCSPModule(decoder=dict(hidden_dim=H, num_layers=L, num_freqs=F, ln=True, edge_style="fc"),
                      beta_scheduler=dict(timesteps=T, scheduler_mode="cosine"),
                      sigma_scheduler=dict(timesteps=T, sigma_begin=0.005, sigma_end=0.5, sigmas_norm=np.load(SIGMAS_NORM)),
                      device=device)
    with torch.no_grad():
        v = m.decoder.views()
        for k in ("coord_out.weight", "lattice_out.weight", "type_out.weight", "type_out.bias"):
            v[k].mul_(HEAD_SCALE)
    m.decoder.mark_dirty()
    return m


def cpu_baseline(budget_s=15.0, na=None, what=None):
    """The CPU oracle (port of the reference's PyTorch path, pinned by tests/golden) on this
    host's cores: a bounded slice of the same workload (B=32 of the 256 crystals, a few
    denoising steps of the 1000), scaled linearly -- per-step cost is t-independent."""
    from oracle import diffcsp_oracle as O
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    P = O.init_params(hp, seed=SEED_W, head_scale=HEAD_SCALE)
    sch = O.Schedules.make(T, sigmas_norm=torch.from_numpy(np.load(SIGMAS_NORM)))
    Bc = 32 if na is None else len(na)
    na = torch.full((Bc,), NATOM, dtype=torch.long) if na is None else torch.tensor(na, dtype=torch.long)
    cores = torch.get_num_threads()
    steps_done, t_total = 0, 0.0
    n_steps = 1
    state = None
    t_cur = T
    while t_total < budget_s and steps_done < 20:
        noise = O.philox_sampler_noise(SEED_NOISE, na, T, t_stop=t_cur - n_steps)
        if state is not None:
            noise["x_T"], noise["l_T"], noise["t_T"] = state
        # run steps t_cur .. t_cur-n_steps+1 by treating them as the head of a chain
        sch_t = sch
        t0 = time.perf_counter()
        final, _ = _oracle_steps(O, P, hp, sch_t, na, noise, t_cur, t_cur - n_steps)
        t_total += time.perf_counter() - t0
        steps_done += n_steps
        t_cur -= n_steps
        state = (final["frac_coords"], final["lattices"], final["atom_types"])
        per = t_total / steps_done
        n_steps = max(1, min(20 - steps_done, int((budget_s - t_total) / per))) if t_total < budget_s else 0
        if n_steps == 0:
            break
    value = Bc * steps_done / (T * t_total)
    return {"value": value, "unit": "structures/s", "cores": cores, "kind": "port",
            "sample": f"oracle/diffcsp_oracle.py (plain torch fp32 CPU), " + (f"B={Bc} crystals x {NATOM} atoms" if what is None else f"{what} ({int(na.sum())} atoms)")
                      + f", {steps_done} of {T} denoising steps in {t_total:.1f} s, scaled linearly (per-step cost is t-independent)"}


def cpu_baseline_ft(budget_s=15.0, na=None, what=None):
    """The CPU oracle's ft_step (restatement of pipeline/mat_invent.py:150-177: noise, agent forward, frozen-prior forward,
    autograd backward) on this host's cores: a bounded slice of the same workload (16 of the 256 crystals, a few timesteps)."""
    from oracle import diffcsp_oracle as O
    hp = O.CSPNetHParams(hidden_dim=H, num_layers=L, num_freqs=F)
    agent, prior = O.init_params(hp, seed=SEED_W, head_scale=HEAD_SCALE), O.init_params(hp, seed=SEED_W, head_scale=HEAD_SCALE)
    sch = O.Schedules.make(T, sigmas_norm=torch.from_numpy(np.load(SIGMAS_NORM)))
    Bc = 16 if na is None else len(na)
    nat = torch.full((Bc,), NATOM, dtype=torch.long) if na is None else torch.tensor(na, dtype=torch.long)
    Nc = int(nat.sum())
    g = torch.Generator().manual_seed(7)
    batch = dict(num_atoms=nat, frac_coords=torch.rand(Nc, 3, generator=g),
                 atom_types=torch.randint(1, 95, (Nc,), generator=g), lengths=4 + 6 * torch.rand(Bc, 3, generator=g),
                 angles=70 + 40 * torch.rand(Bc, 3, generator=g))
    rewards = torch.rand(Bc, generator=g)

    def noise_fn(epoch, t):
        return dict(rand_l=torch.randn(Bc, 3, 3, generator=g), rand_x=torch.randn(Nc, 3, generator=g),
                    rand_t=torch.randn(Nc, 100, generator=g))
    done, t_total, n = 0, 0.0, 1
    while t_total < budget_s and done < 24:
        t0 = time.perf_counter()
        O.ft_step(agent, prior, hp, sch, O.Costs(), batch, rewards, noise_fn, lr=1e-4, timesteps=n, accum_steps=n, sigma=0.025)
        t_total += time.perf_counter() - t0
        done += n
        n = max(1, min(24 - done, int((budget_s - t_total) / (t_total / done))))
    return {"value": Bc * done / t_total, "unit": "crystal-timesteps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/diffcsp_oracle.py ft_step (plain torch fp32 CPU, autograd), " + (f"{Bc} crystals x {NATOM} atoms" if what is None else f"{what} ({Nc} atoms)")
                      + f", {done} timesteps (+ one Adam step per call) in {t_total:.1f} s"}


def _oracle_steps(O, P, hp, sch, na, noise, t_from, t_to):
    """Run oracle steps t_from .. t_to+1 starting from the state in noise[x_T,l_T,t_T]."""
    # O.sample always starts at sch.timesteps; emulate a mid-chain start by a shallow schedule view
    class _S:
        pass
    s = _S()
    s.timesteps, s.beta, s.sigma, s.sigma_begin, s.sigma_end = t_from, sch.beta, sch.sigma, sch.sigma_begin, sch.sigma_end
    return O.sample(P, hp, s, na, noise, step_lr=STEP_LR, t_stop=t_to, keep_traj=False)


def _apply_env_knobs(lib):
    """Experiment knobs (A/B runs of kernel choices); the defaults are what the library ships with."""
    for env, fn in (("MI_DB_MIN_TILES", lib.mi_debug_set_db_min_tiles), ("MI_NODE_PLANES_MIN_ROWS", lib.mi_debug_set_node_planes_min_rows),
                    ("MI_PLANES_SMALL_TILES", lib.mi_debug_set_planes_small_tiles), ("MI_TN128", lib.mi_debug_set_tn128), ("MI_TN_SPLIT_MIN_ROWS", lib.mi_debug_set_tn_split_min_rows), ("MI_TN_TILES", lib.mi_debug_set_tn_target_tiles),
                    ("MI_EDGE_PAIRS", lib.mi_set_edge_pairs), ("MI_PLANES_DMA", lib.mi_debug_set_planes_dma), ("MI_PLANES_BIG_SEG", lib.mi_debug_set_planes_big_seg), ("MI_NODE_PRIORITY", lib.mi_debug_set_node_priority), ("MI_PLANES_LATENCY", lib.mi_debug_set_planes_latency), ("MI_NODE_FUSED", lib.mi_debug_set_node_fused), ("MI_NODE_TRAIN", lib.mi_debug_set_node_train), ("MI_NODE_SPLIT", lib.mi_debug_set_node_split), ("MI_NODE_COLS", lib.mi_debug_set_node_cols), ("MI_NODE_TOUCH", lib.mi_debug_set_node_touch), ("MI_HEADS_ROWS16", lib.mi_debug_set_heads_rows16), ("MI_EVAL_REUSE", lib.mi_debug_set_eval_reuse), ("MI_SKIP", lib.mi_debug_set_skip), ("MI_RT_LEAN", lib.mi_debug_set_rt_lean), ("MI_EDGE2_FUSED", lib.mi_debug_set_edge2_fused), ("MI_EDGE1_FUSED", lib.mi_debug_set_edge1_fused), ("MI_EDGE_FUSED", lib.mi_debug_set_edge_fused), ("MI_MG_NOSYNC", lib.mi_debug_set_mg_nosync)):
        if os.environ.get(env) is not None and os.environ[env] != "":
            fn(int(os.environ[env]))
    if os.environ.get("MI_NODE_BWD", "") != "":   # the fused node-level backward chain of the fine-tune step: 1 (default) / 0 = seven launches per layer
        lib.mi_debug_set_node_bwd(int(os.environ["MI_NODE_BWD"]), int(os.environ.get("MI_NODE_BWD_MIN_BLOCKS", "0")))
    if os.environ.get("MI_PLANES_RT", "") != "":   # register-tile form of the large plane products: 0 off, 1 those with epilogue extensions (default), 2 all
        lib.mi_debug_set_planes_rt(int(os.environ["MI_PLANES_RT"]), 0)
    if os.environ.get("MI_FT_PRIOR_AUX", "") != "":
        from matinvent_amd import finetune
        finetune.FT_PRIOR_AUX = bool(int(os.environ["MI_FT_PRIOR_AUX"]))
    if os.environ.get("MI_WGRAD_WINDOW", "") != "":   # micro-steps per node-level weight-gradient contraction of the fine-tune loop (0: off)
        from matinvent_amd import finetune
        finetune.WGRAD_WINDOW = int(os.environ["MI_WGRAD_WINDOW"])


def measure_ft(args, K, W, ctx, cpu_budget_s=15.0):
    """Secondary metric (BASELINE configs[2]/[3]): crystal-timesteps / second of the fine-tune loop
    (noise + agent fwd + frozen-prior fwd + agent bwd per timestep, fused Adam every 50 -- and after the last timestep of a call, as
    pipeline/mat_invent.py:176-177 does), ft set = 256 synthetic crystals x 20 atoms per GPU, reward ~ U[0,1] (stands in for reward=hhi).
    Returns the bench line (rank 0) or None."""
    world, rank, local_rank, share, dist = ctx
    dev = torch.device("cuda", local_rank)
    from matinvent_amd.data import CrystalData
    from matinvent_amd.finetune import ft_step
    from matinvent_amd import _lib
    _apply_env_knobs(_lib.load())
    agent, prior = build_module(dev), build_module(dev)
    prior.requires_grad_(False)
    g = torch.Generator().manual_seed(7)
    nglob = B * world
    data = [CrystalData(torch.rand(NATOM, 3, generator=g), torch.randint(1, 95, (NATOM,), generator=g), 4 + 6 * torch.rand(1, 3, generator=g),
                        70 + 40 * torch.rand(1, 3, generator=g)) for _ in range(nglob)]
    rewards = torch.rand(nglob, generator=g).numpy()
    cfg = dict(lr=1e-4, accum_steps=50, epochs=1, sigma=0.025)

    def run(n):
        ft_step(agent, prior, data, rewards, dict(cfg, timesteps=n), log=lambda *_: None, groups=getattr(args, "ft_groups", None))
        dist_barrier(ctx)

    import ctypes as C
    lib = _lib.load()
    run(max(W, 1))
    if getattr(args, "counter_child", False):   # (under rocprofv3 --pmc: the dispatches of K micro-steps are what is wanted, nothing is printed)
        run(K)
        return None
    _lib.check(lib.mi_profile_enable(agent.decoder._h, 1))
    f16, f32 = C.c_double(), C.c_double()
    _lib.check(lib.mi_debug_mfma_flops(None, None, 1))   # (host-side counters of the matrix-pipe work every launcher issues: reset)
    t0 = time.perf_counter()
    run(K)
    elapsed = dist_max_time(ctx, time.perf_counter() - t0)
    _lib.check(lib.mi_debug_mfma_flops(C.byref(f16), C.byref(f32), 0))
    n_launch, tot_ms, union_ms = C.c_int64(), C.c_double(), C.c_double()
    _lib.check(lib.mi_profile_read(agent.decoder._h, C.byref(n_launch), C.byref(tot_ms), C.byref(union_ms)))
    _lib.check(lib.mi_profile_enable(agent.decoder._h, 0))
    allreduce_ms = None
    if dist is not None:   # the flat-gradient all-reduce on a buffer of the gradient's size (49.4 MB), timed on its own after the region
        from matinvent_amd.dist import allreduce_flat_
        gbuf = torch.zeros_like(agent.decoder.theta)
        allreduce_flat_(gbuf)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(5):
            allreduce_flat_(gbuf)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - ta) / 5 * 1e3
        del gbuf
    del agent, prior
    if rank != 0:
        return None
    flops = 4 * 5.893e9 * nglob * K  # agent fwd + prior fwd + 2x for backward
    # dominant kernel pair of the micro-step (profiles/*_finetune.md): the agent's forward edge stage -- the same two plane GEMMs
    # as the sampler's, bracketed by HIP events on their launch stream; one launch = one layer over one crystal group
    groups = max(1, n_launch.value // max(1, K * L))
    E = B * NATOM * NATOM / groups
    f_exec, f_alg = edge_flops_per_edge(pairs=True)
    terms = 3 if lib.mi_plane_format() == 2 else 6
    busy_ms = union_ms.value if groups > 1 else tot_ms.value
    issued = terms * n_launch.value * E * f_exec / (busy_ms * 1e-3) / 1e12
    # the WHOLE micro-step (noise, agent training forward, frozen-prior forward, loss, backward; Adam when it falls due): matrix-pipe flops
    # ISSUED by every product of the timed region (the launchers count 2 M N K x MFMA terms as they enqueue: mi_debug_mfma_flops) / elapsed
    whole16 = f16.value / elapsed / 1e12
    whole32 = f32.value / elapsed / 1e12
    traffic, traffic_src = None, {"measured": "skipped (--no-counters)" if getattr(args, "no_counters", False) else "not applicable to this world size"}
    if world == 1 and not getattr(args, "no_counters", False):
        traffic, traffic_src = measure_ft_traffic_live(args)
    out = {"metric": "fine-tune crystal-timesteps/sec", "value": nglob * K / elapsed, "unit": "crystal-timesteps/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True,
           "scaling": getattr(args, "scaling", "weak"), "vs_baseline": None,
           "dtype": "f32 (forward and edge-level backward products: 2-plane fp16 split, 3 MFMA terms; node-level backward products: 3-plane bf16 split, 6 terms; f32 accumulate)",
           "data": "synthetic",
           "config": {"workload": "BASELINE configs[2]: mat_invent fine-tune micro-steps (noise + agent fwd + frozen-prior fwd + agent bwd per "
                                  f"timestep), {getattr(args, 'batch_label', '256 crystals per GPU')} x 20 atoms, synthetic reward, accum_steps=50, fused Adam, one flat-gradient "
                                  "all-reduce (RCCL) per optimizer step when N>1; a bench step = one timestep over the batch",
                      "batch_per_gpu": B, "atoms_per_cell": NATOM, "accum_steps": 50, "concurrent_groups": groups,
                      "adam_steps_in_timed_region": K // 50 + (1 if K % 50 else 0),
                      "comm_backend": (dist.get_backend() if dist is not None else None), "world_size": world,
                      "flat_gradient_allreduce_ms": allreduce_ms},
           "roofline": {"bound": "mfma", "scope": "whole micro-step: every matrix product of noise + agent training forward + frozen-prior forward + loss + backward "
                                                  "(+ Adam when due) over the timed region's elapsed time; one 'launch' = one timestep over the batch",
                        "kernel": "all products of a micro-step (edge_gemm1b / edge_gemm2b / node_chain forward, gemm_rt dM1, gemm_tn_planes dW, node-level backward products)",
                        "achieved": whole16, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": whole16 / PEAK_BF16_MFMA_TFLOPS,
                        "traffic": traffic, "traffic_source": traffic_src, "launches": K, "avg_launch_ms": elapsed * 1e3 / K,
                        "flops_per_launch_issued_16bit_pipe": f16.value / K, "flops_per_launch_f32_mfma": f32.value / K,
                        "f32_mfma_achieved_tflops": whole32, "f32_mfma_frac_of_its_peak": whole32 / PEAK_F32_MFMA_TFLOPS,
                        "algorithmic_section8d_flops_per_launch": 4 * 5.893e9 * nglob,
                        "note": "achieved = 16-bit-pipe MFMA flops issued (3 terms per fp32 product on pre-split fp16 planes, 6 on bf16 planes split on the fly) / "
                                "elapsed; the f32-input MFMA products (short weight-gradient contractions) are priced separately against their own 157.3 TF/s peak; "
                                "traffic = FETCH_SIZE x2 + WRITE_SIZE of ALL kernels per timestep",
                        "edge_stage_forward": {"kernel": "edge_gemm1b_kernel + edge_gemm2b_kernel (agent forward, edge MLP of one layer), HIP-event bracketed",
                                               "achieved": issued, "frac": issued / PEAK_BF16_MFMA_TFLOPS, "launches": int(n_launch.value),
                                               "avg_launch_ms": tot_ms.value / max(1, n_launch.value), "concurrent_streams": groups, "stage_busy_ms": busy_ms,
                                               "flops_per_launch_executed": E * f_exec, "flops_per_launch_section8d": E * f_alg}},
           "end_to_end": {"tflops_section8d": flops / elapsed / 1e12,
                          "frac_of_f32_mfma_peak_section8d": flops / elapsed / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)}}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_ft(cpu_budget_s)
    return out


def main_ft(args):
    K, W = (args.steps if args.steps != 1000 else 100), args.warmup
    ctx = dist_setup(args.gpus)
    out = measure_ft(args, K, W, ctx)
    if out is not None:
        print(json.dumps(out), flush=True)
    if ctx[4] is not None:
        dist_barrier(ctx)
        ctx[4].destroy_process_group()


def _edge_stage_roofline(lib, module, edges_total, pairs_total, evals):
    """Event-bracketed edge stage (the two plane GEMMs of a layer, mi_profile_*) of a run over a RAGGED set: matrix-pipe flops issued by all
    bracketed launches / the union of their execution intervals.  edges_total / pairs_total: directed edges (self edges included) and
    unordered atom pairs of the whole set; evals: network evaluations of the profiled module in the timed region."""
    import ctypes as C
    from matinvent_amd import _lib
    n_launch, tot_ms, union_ms = C.c_int64(), C.c_double(), C.c_double()
    _lib.check(lib.mi_profile_read(module.decoder._h, C.byref(n_launch), C.byref(tot_ms), C.byref(union_ms)))
    _lib.check(lib.mi_profile_enable(module.decoder._h, 0))
    terms = 3 if lib.mi_plane_format() == 2 else 6
    fp32_flops = evals * L * (pairs_total * 2 * (6 * F) * H + edges_total * 2 * H * H)   # Fourier block over pairs + second linear over edges
    busy_ms = max(union_ms.value, 1e-9)
    issued = terms * fp32_flops / (busy_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "edge_gemm1b_kernel (pair mode; gemm_planes_kernel<pair> for small launches) + edge_gemm2b_kernel (edge MLP of one layer over one crystal group; the second GEMM on the register-tile kernel in inference forwards at hidden_dim 512)", "achieved": issued,
            "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": issued / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "launches": int(n_launch.value),
            "avg_launch_ms": tot_ms.value / max(1, n_launch.value), "stage_busy_ms": busy_ms, "achieved_fp32_equivalent": issued / terms,
            "note": "flops of all bracketed launches / union of their execution intervals (concurrent groups overlap); small ragged sets are "
                    "latency-bound launches of at most one workgroup per CU, so the fraction is low by construction"}


def main_reference_defaults(args):
    """Secondary lines on the reference's REAL default workloads (not the headline):
      sample-default: ragged mp_20 atom counts (models/diffcsp/sample.py:42-62, numpy seed 0), B = 192 = eval_size x 12
                      (configs/model/diffcsp.yaml:7), T = 1000, automatic concurrent chains;
      ft-default:     the default fine-tune set of 18 crystals = top-k 8 + replay <= 10 (configs/pipeline/mat_invent.yaml), stacked
                      timesteps (the set is bound by the host's launch rate otherwise), accum_steps 50, fused Adam."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from matinvent_amd import _lib
    from matinvent_amd.sampling import ATOM_DIST
    lib = _lib.load()
    _apply_env_knobs(lib)
    np.random.seed(0)
    p = ATOM_DIST["mp_20"]
    if args.mode == "sample-default":
        Bd = 192
        na = np.random.choice(len(p), Bd, p=p)
        K, W = (args.steps if args.steps != 1000 else 50), max(1, args.warmup)
        m = build_module(dev)

        class Counts:
            num_atoms = torch.tensor(na)
        cb = Counts()
        m.sample(cb, seed=SEED_NOISE + 1, step_lr=STEP_LR, t_start=T, t_stop=T - W)
        final, _ = m.sample(cb, seed=SEED_NOISE, step_lr=STEP_LR, t_start=T, t_stop=T)
        state = (final["frac_coords"], final["lattices"], final["atom_types"])
        torch.cuda.synchronize()
        _lib.check(lib.mi_profile_enable(m.decoder._h, 1))
        t0 = time.perf_counter()
        m.sample(cb, seed=SEED_NOISE, step_lr=STEP_LR, init=state, t_start=T, t_stop=T - K)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        roof = _edge_stage_roofline(lib, m, int((na.astype(np.int64) ** 2).sum()), int((na.astype(np.int64) * (na - 1) // 2).sum()), 2 * K)
        out = {"metric": "crystal structures/sec (1000-step reverse diffusion), reference default sampling batch", "value": Bd * K / (T * elapsed),
               "unit": "structures/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32 via 2-plane fp16 split", "data": "synthetic",
               "config": {"workload": "reference default: 192 crystals with mp_20 atom counts (1..20, numpy seed 0), T=1000, DiffCSP CSPNet H=512 L=6 F=128",
                          "batch_per_gpu": Bd, "atoms_total": int(na.sum()), "edges_total": int((na.astype(np.int64) ** 2).sum())},
               "roofline": roof}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(na=[int(v) for v in na[:48]], what="the first 48 of the 192 ragged crystals")
    else:
        from matinvent_amd.data import CrystalData
        from matinvent_amd.finetune import ft_step
        nset = 18
        na = np.random.choice(len(p), nset, p=p)
        K, W = (args.steps if args.steps != 1000 else 100), max(1, args.warmup)
        agent, prior = build_module(dev), build_module(dev)
        prior.requires_grad_(False)
        g = torch.Generator().manual_seed(7)
        data = [CrystalData(torch.rand(int(n), 3, generator=g), torch.randint(1, 95, (int(n),), generator=g), 4 + 6 * torch.rand(1, 3, generator=g),
                            70 + 40 * torch.rand(1, 3, generator=g)) for n in na]
        rewards = torch.rand(nset, generator=g).numpy()
        cfg = dict(lr=1e-4, accum_steps=50, epochs=1, sigma=0.025)
        ft_step(agent, prior, data, rewards, dict(cfg, timesteps=max(W, 50)), log=lambda *_: None)
        torch.cuda.synchronize()
        _lib.check(lib.mi_profile_enable(agent.decoder._h, 1))
        t0 = time.perf_counter()
        ft_step(agent, prior, data, rewards, dict(cfg, timesteps=K), log=lambda *_: None)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        roof = _edge_stage_roofline(lib, agent, int((na.astype(np.int64) ** 2).sum()), int((na.astype(np.int64) * (na - 1) // 2).sum()), K)
        roof["kernel"] += " -- the agent's training forward"
        out = {"metric": "fine-tune crystal-timesteps/sec, reference default fine-tune set", "value": nset * K / elapsed, "unit": "crystal-timesteps/s",
               "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32 via 2-plane fp16 / 3-plane bf16 splits", "data": "synthetic",
               "config": {"workload": "reference default fine-tune set: 18 crystals with mp_20 atom counts (top-k 8 + replay 10), accum_steps 50, stacked "
                                      "timesteps (automatic), fused Adam; an RL step runs 3 x 1000 such timesteps",
                          "set_size": nset, "atoms_total": int(na.sum()), "edges_total": int((na.astype(np.int64) ** 2).sum()),
                          "seconds_per_rl_step_finetune": 3000 * elapsed / K},
               "roofline": roof}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_ft(na=[int(v) for v in na], what="the same 18 ragged crystals")
    print(json.dumps(out), flush=True)


def cpu_baseline_mg(budget_s=20.0):
    """oracle/mattergen_oracle.py (plain torch fp32 CPU) on this host's cores: predictor-corrector steps of the same network on a bounded
    slice (2 of the 256 crystals, a few of the 1000 grid points from the same mid-chain state), scaled linearly."""
    from oracle import mattergen_oracle as MO
    hp = MO.GemNetHParams()
    P = MO.init_params(hp, seed=SEED_W, head_scale=20.0)
    Bc = 2
    na = torch.full((Bc,), NATOM, dtype=torch.long)
    N = Bc * NATOM
    g = torch.Generator().manual_seed(3)
    mu = (NATOM / 0.05771451654022283) ** (1 / 3)
    cell = mu * torch.eye(3)[None].repeat(Bc, 1, 1) + 0.3 * torch.randn(Bc, 3, 3, generator=g)
    state = dict(pos=torch.rand(N, 3, generator=g), cell=0.5 * (cell + cell.transpose(1, 2)), atomic_numbers=torch.randint(1, 101, (N,), generator=g))
    done, t_total, i0 = 0, 0.0, 500
    with torch.no_grad():
        while t_total < budget_s and done < 8:
            nz = dict(init_pos=None, init_cell=None)
            for k, shp, fn in (("corr_pos", (N, 3), torch.randn), ("corr_cell", (Bc, 3, 3), torch.randn), ("pred_pos", (N, 3), torch.randn),
                               ("pred_cell", (Bc, 3, 3), torch.randn), ("pred_u1", (N,), torch.rand), ("pred_u2", (N,), torch.rand)):
                nz[k] = [None] * (i0 + done) + [fn(*shp, generator=g)]
            t0 = time.perf_counter()
            s, _ = MO.pc_sample(P, hp, MO.Corruption(), na, nz, n_steps=T, t_stop_index=i0 + done + 1, start_index=i0 + done, state=state)
            t_total += time.perf_counter() - t0
            state = dict(pos=s["pos"], cell=s["cell"], atomic_numbers=s["atomic_numbers"])
            done += 1
    return {"value": Bc * done / (T * t_total), "unit": "structures/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/mattergen_oracle.py (plain torch fp32 CPU; graph and triplet sums are Python loops), {Bc} crystals x {NATOM} atoms, {done} of {T} "
                      f"predictor-corrector steps in {t_total:.1f} s, scaled linearly"}


def cpu_baseline_mg_ft(budget_s=20.0):
    """oracle/mattergen_oracle.py fine-tune timesteps (noising, agent forward, frozen-prior forward, torch autograd backward) on this host's
    cores: a bounded slice of the same workload (4 of the crystals, one or two timesteps)."""
    from oracle import mattergen_oracle as MO
    hp = MO.GemNetHParams()
    A = {k: v.requires_grad_(True) for k, v in MO.init_params(hp, seed=SEED_W, head_scale=20.0).items()}
    Q = MO.init_params(hp, seed=SEED_W, head_scale=20.0)
    Bc = 4
    g = torch.Generator().manual_seed(7)
    mu = (NATOM / 0.05771451654022283) ** (1 / 3)
    na = torch.full((Bc,), NATOM, dtype=torch.long)
    cell = mu * torch.eye(3)[None].repeat(Bc, 1, 1) + 0.3 * MO.symmetric_noise(torch.randn(Bc, 3, 3, generator=g))
    ob = dict(pos=torch.rand(Bc * NATOM, 3, generator=g), cell=cell, atomic_numbers=torch.randint(1, 95, (Bc * NATOM,), generator=g), num_atoms=na)
    rw = torch.rand(Bc, generator=g)
    corr = MO.Corruption()
    done, t_total = 0, 0.0
    while t_total < budget_s and done < 4:
        nz = dict(pos=torch.randn(Bc * NATOM, 3, generator=g), cell=torch.randn(Bc, 3, 3, generator=g), types=torch.rand(Bc * NATOM, generator=g))
        t0 = time.perf_counter()
        t = torch.full((Bc,), MO.time_grid(corr, 500 + done))
        noisy, aux = MO.sample_marginal(corr, ob, t, nz)
        pa = MO.gemnet_forward(A, hp, noisy["pos"], noisy["cell"], noisy["atomic_numbers"], na, t)
        with torch.no_grad():
            pp = MO.gemnet_forward(Q, hp, noisy["pos"], noisy["cell"], noisy["atomic_numbers"], na, t)
        sl, _ = MO.sample_loss(corr, ob, aux, pa)
        kl = MO.calc_kl_reg(pa, pp, aux["node2graph"], Bc)
        MO.param_grads((rw * sl + 0.025 * kl * (1.1 - rw)).mean(), A)
        t_total += time.perf_counter() - t0
        done += 1
    return {"value": Bc * done / t_total, "unit": "crystal-timesteps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/mattergen_oracle.py fine-tune timesteps (plain torch fp32 CPU, autograd; graph and triplet sums are Python loops), {Bc} crystals x "
                      f"{NATOM} atoms, {done} timesteps in {t_total:.1f} s"}


def measure_mg(args, K, W, ctx=None):
    """The MatterGen-LABELLED form of BASELINE configs[1]: the predictor-corrector reverse sampler of the MatterGen-shaped network
    (GemNet-T shape: 4 blocks at 512 / 512 / 64 / 16 / 16, cutoff 7 A, <= 50 neighbours, triplet basis; 42.7 M parameters), batch 256 x
    20 atoms, 1000-point grid, two denoiser evaluations per step.  SELF-CONSISTENT, PARITY-UNPINNED vs upstream (the reference's
    MatterGen arithmetic is an un-vendored dependency).  Returns the fields of a bench line."""
    ctx = ctx or (1, 0, torch.cuda.current_device(), False, None)
    world, rank, local_rank, share, dist = ctx
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    from matinvent_amd import _lib
    from matinvent_amd.mattergen import MatterGenModule
    lib = _lib.load()
    if os.environ.get("MI_MG_LEAN"):   # ablation of the lean-inference switches (scripts/gpu_mg_ab.sh); the raw bit mask
        _lib.check(lib.mi_debug_set_mg_lean(-int(os.environ["MI_MG_LEAN"])))
    torch.manual_seed(SEED_W)
    m = MatterGenModule(device=dev)
    # random-init heads scaled so that score x std is of order one (a trained denoiser's range): the signal-to-noise Langevin step
    # is 2 (snr |z| / |score|)^2, and a network with tiny outputs would random-walk the cells out of the cutoff within a few steps
    m.decoder.reset_parameters(head_scale=20.0, cell_head_scale=400.0)
    Bm = args.mg_batch
    na = [NATOM] * Bm
    # a mid-chain state (t ~ 0.5): cells around the limit mean, the regime most of the chain runs in
    g = torch.Generator().manual_seed(3)
    N = Bm * NATOM
    mu = (NATOM / 0.05771451654022283) ** (1 / 3)
    state = dict(pos=torch.rand(N, 3, generator=g), cell=mu * torch.eye(3)[None].repeat(Bm, 1, 1) + 0.3 * torch.randn(Bm, 3, 3, generator=g),
                 atomic_numbers=torch.randint(1, 101, (N,), generator=g))
    state["cell"] = 0.5 * (state["cell"] + state["cell"].transpose(1, 2))
    i0 = 500
    E0 = int(m._batch_for(torch.tensor(na)).graph(state["pos"], state["cell"])["src"].shape[0])
    chains = max(1, int(getattr(args, "mg_chains", 4)))
    okw = dict(node_offset=rank * N, graph_offset=rank * Bm)   # global ids: every rank samples its own crystals of one global batch (weak scaling)
    s, _ = m.sample(na, n_steps=T, seed=SEED_NOISE, i_start=i0, i_stop=i0 + W, state=state, chains=chains, **okw)
    st = dict(pos=s["pos"], cell=s["cell"], atomic_numbers=s["atomic_numbers"])
    dist_barrier(ctx)
    # A random-init denoiser cannot hold a crystal together: left to itself the chain inflates the cells within a few steps and the edge
    # count falls (256k -> 60k over ten steps, nothing left after a few hundred), so a free-running chain measures an emptying graph.  A
    # trained model keeps cells near physical densities, where every atom has its 50 neighbours inside the cutoff -- the 256k-edge regime
    # of the starting state.  Default: every timed step starts from that state (one-step calls; all kernels of a step run, only the
    # hand-over of the state to the next step is replaced); --mg-free-chain times the free-running chain instead.
    hold = not getattr(args, "mg_free_chain", False)
    t0 = time.perf_counter()
    if hold:
        for k in range(K):
            s, mean = m.sample(na, n_steps=T, seed=SEED_NOISE + k, i_start=i0 + W + k, i_stop=i0 + W + k + 1, state=st, chains=chains, **okw)
    else:
        s, mean = m.sample(na, n_steps=T, seed=SEED_NOISE, i_start=i0 + W, i_stop=i0 + W + K, state=st, chains=chains, **okw)
    dist_barrier(ctx)
    elapsed = dist_max_time(ctx, time.perf_counter() - t0)
    if getattr(args, "counter_child", False):
        return None
    gr = m._batch_for(torch.tensor(na)).graph(s["pos"], s["cell"])
    E = int(gr["src"].shape[0])
    finite = bool(torch.isfinite(mean["pos"]).all()) and bool(torch.isfinite(mean["cell"]).all())
    hp = m.decoder.hp
    Em = 0.5 * (E0 + E)
    flops_eval = mg_flops_eval(hp, N, Em)
    terms = 3 if lib.mi_plane_format() == 2 else 6
    sat = _lib.saturation_events(reset=True)
    nparams = int(m.decoder.theta.numel())
    del m
    # north_star: "achieved HBM GB/s against the chip's peak" for this sampler.  `frac` prices the ALGORITHMIC bytes (Y_eval_mg above: what
    # a perfectly fused implementation has to move) against the 8 TB/s peak over the measured time; beside it the bytes the kernels
    # actually moved (FETCH_SIZE x2 + WRITE_SIZE of the committed rocprofv3 passes of this command, scripts/gpu_mg_prof.sh -- bench.py cannot
    # read hardware counters itself) as a ratio to the algorithmic ones.  The network is matrix-pipe-bound by nature (F / Y >> the chip's
    # balance point), so a low algorithmic fraction is a property of the workload; the ratio is what measures wasted traffic.
    y_eval = mg_bytes_per_crystal_eval(hp, NATOM, Em / Bm, nparams, Bm)
    alg_gbps = y_eval * 2 * Bm * K * world / elapsed / 1e9
    hbm = {"bound": "hbm", "achieved": alg_gbps, "peak": PEAK_HBM_TBPS * 1e3 * world, "unit": "GB/s", "frac": alg_gbps / (PEAK_HBM_TBPS * 1e3 * world),
           "algorithmic_bytes_per_crystal_evaluation": y_eval, "flops_per_byte": flops_eval / Bm / y_eval,
           "counter_over_algorithmic": None, "counter_GBps": None}
    if world == 1 and not getattr(args, "no_counters", False) and getattr(args, "mg_live_counters", True):
        step_bytes, src = measure_mg_traffic_live(args)   # (the bytes the kernels of a step actually moved, measured by this run)
        hbm["counter_source"] = src
        if step_bytes is not None:
            hbm["counter_bytes_per_step"] = step_bytes
            hbm["counter_over_algorithmic"] = step_bytes / (2 * Bm * y_eval)
            hbm["counter_GBps"] = step_bytes * K / elapsed / 1e9
            hbm["counter_frac_of_peak"] = hbm["counter_GBps"] / (PEAK_HBM_TBPS * 1e3)
    if rank != 0:
        return None
    return {"metric": "crystal structures/sec (1000-step reverse diffusion), MatterGen-shaped network", "value": world * Bm * K / (T * elapsed), "unit": "structures/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32: edge-level layers on pre-split two-plane fp16 operands (3 MFMA terms), " if terms == 3 else "f32: edge-level layers on pre-split three-plane bf16 operands (6 MFMA terms), ")
                     + "node-level layers on three bf16 planes split on the fly (6 terms); f32 accumulate",
            "data": "synthetic",
            "config": {"workload": f"MatterGen-labelled form of BASELINE configs[1]: predictor-corrector sampler of the MatterGen-shaped network, batch={Bm} "
                                   "crystals x 20 atoms, 2 denoiser evals/step, "
                                   + ("every timed step from the same mid-chain state at physical density (50 neighbours per atom: the regime a trained "
                                      "model keeps; a random-init chain left to itself inflates its cells and empties its graph)" if hold else
                                      "free-running random-init chain (its cells drift, so the edge count falls during the run: compare lines of equal steps / warmup)")
                                   + "; SELF-CONSISTENT, PARITY-UNPINNED vs upstream",
                       "state": "held" if hold else "free-running",
                       "batch_per_gpu": Bm, "atoms_per_cell": NATOM, "T": T, "concurrent_chains": chains, "edges_first_step": E0, "edges_last_step": E,
                       "comm_backend": (dist.get_backend() if dist is not None else None), "world_size": world,
                       "parameters": nparams, "final_state_finite": finite, "fp16_plane_saturation_events": sat},
            "roofline": {"bound": "mfma", "kernel": "gemm_rt_kernel<lean> (edge-level dense layers of the interaction / output blocks)",
                         "achieved": terms * flops_eval * 2 * K / elapsed / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": terms * flops_eval * 2 * K / elapsed / 1e12 / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "per_gpu": True,
                         "achieved_fp32_equivalent": flops_eval * 2 * K / elapsed / 1e12, "flops_per_evaluation": flops_eval,
                         "note": "end-to-end rate of the dense-layer flops (whole step time, all kernels); per-kernel durations and HBM GB/s: profiles/"},
            "hbm_roofline": hbm}


def main_mg(args):
    """Secondary line: the MatterGen-shaped sampler (see measure_mg); the headline `value` stays on the pinned DiffCSP network."""
    K, W = (args.steps if args.steps != 1000 else 10), max(1, min(args.warmup, 2))
    ctx = dist_setup(args.gpus)
    _apply_env_knobs(__import__("matinvent_amd._lib", fromlist=["load"]).load())
    out = measure_mg(args, K, W, ctx)
    if out is not None:
        if not args.no_cpu_baseline and ctx[0] == 1:
            out["cpu_baseline"] = cpu_baseline_mg()
        print(json.dumps(out), flush=True)
    if ctx[4] is not None:
        dist_barrier(ctx)
        ctx[4].destroy_process_group()


def main_mg_ft(args):
    """Secondary line, MatterGen-LABELLED form of BASELINE configs[2]: fine-tune timesteps of the MatterGen-shaped network through the
    reference's loop (add_noise, agent forward, frozen-prior forward, tape backward, fused Adam every accum_steps).  Self-consistent,
    parity-unpinned vs upstream."""
    K, W = (args.steps if args.steps != 1000 else 6), 1
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from matinvent_amd import _lib
    from matinvent_amd.finetune import ft_step
    from matinvent_amd.mattergen import ChemGraph, MatterGenModule
    _lib.load()
    torch.manual_seed(SEED_W)
    agent, prior = MatterGenModule(device=dev), MatterGenModule(device=dev)
    agent.decoder.reset_parameters(head_scale=20.0)
    prior.decoder.load_state_dict(agent.decoder.state_dict())
    prior.requires_grad_(False)
    Bm = args.mg_batch
    g = torch.Generator().manual_seed(7)
    mu = (NATOM / 0.05771451654022283) ** (1 / 3)
    data = [ChemGraph(torch.rand(NATOM, 3, generator=g), mu * torch.eye(3)[None] + 0.3 * torch.randn(1, 3, 3, generator=g),
                      torch.randint(1, 95, (NATOM,), generator=g)) for _ in range(Bm)]
    rewards = torch.rand(Bm, generator=g).numpy()
    cfg = dict(lr=1e-5, accum_steps=50, epochs=1, sigma=0.025)
    ft_step(agent, prior, data, rewards, dict(cfg, timesteps=W), log=lambda *_: None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ft_step(agent, prior, data, rewards, dict(cfg, timesteps=K), log=lambda *_: None)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    out = {"metric": "fine-tune crystal-timesteps/sec, MatterGen-shaped network", "value": Bm * K / elapsed, "unit": "crystal-timesteps/s", "n_gpus": 1,
           "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": ("f32: edge-level layers, their data gradients and weight gradients on two fp16 planes (3 MFMA terms); node-level products on three "
                     "bf16 planes split on the fly (6 terms); f32 accumulate") if _lib.load().mi_plane_format() == 2 else "f32 via 3-plane bf16 split (6 MFMA terms)",
           "data": "synthetic",
           "config": {"workload": f"MatterGen-labelled form of BASELINE configs[2]: fine-tune timesteps (noise, agent fwd, frozen-prior fwd, backward; Adam every 50), "
                                  f"{Bm} crystals x 20 atoms, synthetic reward; SELF-CONSISTENT, PARITY-UNPINNED vs upstream",
                      "batch_per_gpu": Bm, "parameters": int(agent.decoder.theta.numel()),
                      "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9}}
    # dense-layer flops of a timestep = agent forward + frozen-prior forward + backward (data and weight gradients: 2x a forward), at the
    # set's edge count; the forward's products issue 3 fp16 MFMA terms each, the backward's 3 (edge level) or 6 (node level): priced at 3
    # (a lower bound on the matrix-pipe work actually issued)
    gb = agent._batch_for(torch.tensor([NATOM] * min(Bm, 64)))
    d0 = data[:min(Bm, 64)]
    c64 = torch.cat([d.cell for d in d0])
    E64 = int(gb.graph(torch.cat([d.pos for d in d0]), 0.5 * (c64 + c64.transpose(1, 2)))["src"].shape[0])   # (the dataset transform symmetrises the cells)
    Eset = E64 * Bm / min(Bm, 64)
    f_eval = mg_flops_eval(agent.decoder.hp, Bm * NATOM, Eset)
    terms = 3 if _lib.load().mi_plane_format() == 2 else 6
    out["roofline"] = {"bound": "mfma", "kernel": "gemm_rt_kernel / gemm_planes_kernel<0, 2> (edge-level dense layers: forward, data gradients) + gemm_tn_split_kernel (weight gradients)",
                       "achieved": terms * 4 * f_eval * K / elapsed / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                       "frac": terms * 4 * f_eval * K / elapsed / 1e12 / PEAK_BF16_MFMA_TFLOPS, "traffic": None,
                       "achieved_fp32_equivalent": 4 * f_eval * K / elapsed / 1e12, "flops_per_evaluation": f_eval, "edges": Eset,
                       "note": "end-to-end rate of the dense-layer flops of a timestep (2 forwards + backward = 4 x F_eval_mg) over the whole step time"}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_mg_ft()
    print(json.dumps(out), flush=True)


def _child_env(**kw):
    """Environment of a single-process child run of this file (counter passes, the TF32-class leg): never the parent's rendezvous."""
    env = {k: v for k, v in os.environ.items() if k not in ("MI_BENCH_FORCE_DIST", "MI_DIST_FORCE_COLLECTIVES", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT",
                                                            "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(kw)
    return env


def _counter_passes(extra_argv, timeout_s=240):
    """FETCH_SIZE and WRITE_SIZE of every kernel of `python bench.py <extra_argv> --counter-child`, from two rocprofv3 child runs (the
    counters in separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Returns ({counter: {kernel: (dispatches,
    average KiB)}}, None) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "no: rocprofv3 not found on this box"
    here = os.path.abspath(__file__)
    got = {}
    tmp = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "-o", "r", "--", sys.executable, here] + list(extra_argv) + ["--no-cpu-baseline", "--no-counters", "--counter-child"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=_child_env(TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"no: the {counter} pass exceeded {timeout_s} s"
            dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"no: the {counter} pass failed (rc {r.returncode}): {(r.stderr or '')[-200:]}"
            cur = sqlite3.connect(dbs[0]).cursor()
            got[counter] = {k.split("(")[0]: (n, v) for k, n, v in
                            cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,))}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got, None


COUNTER_CORRECTION = "KiB -> bytes; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported"


def measure_traffic_live(args, steps=3, warmup=1):
    """roofline.traffic of the headline line: FETCH_SIZE and WRITE_SIZE of the edge stage's two kernels (pair-mode Fourier GEMM +
    second-linear GEMM) per dispatch, measured by THIS run (see _counter_passes; `steps` denoising steps of the same workload).
    Returns (bytes per bench launch or None, provenance dict)."""
    t0 = time.perf_counter()
    got, why = _counter_passes(["--steps", str(steps), "--warmup", str(warmup), "--streams", str(args.streams), "--path", args.path])
    if got is None:
        return None, {"measured": why}
    stage = lambda k: ("gemm_planes_kernel<1" in k) or ("edge_gemm2" in k) or ("edge_gemm1" in k) or ("edge_ring" in k) or ("edge_fused" in k)
    kernels, total = {}, 0.0
    for k in sorted(set(got["FETCH_SIZE"]) | set(got["WRITE_SIZE"])):
        if not stage(k):
            continue
        nf, f = got["FETCH_SIZE"].get(k, (0, 0.0))
        nw, w = got["WRITE_SIZE"].get(k, (0, 0.0))
        fb, wb = 2.0 * f * 1024.0, w * 1024.0
        kernels[k[:80]] = {"dispatches": nf, "fetch_bytes": fb, "write_bytes": wb}
        total += fb + wb   # (one dispatch of each kernel per bench launch = the edge stage of one layer)
    if not kernels:
        return None, {"measured": "no: the edge-stage kernels do not appear in the counter passes"}
    return total, {"measured": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this command line, after the timed region",
                   "steps_per_pass": steps, "per_kernel": kernels, "wall_s": round(time.perf_counter() - t0, 1), "correction": COUNTER_CORRECTION}


def measure_ft_traffic_live(args, steps=4, warmup=1):
    """roofline.traffic of the fine-tune line: HBM-side bytes of ALL kernels of a micro-step (FETCH_SIZE x2 + WRITE_SIZE over every dispatch of
    a child run of `--mode ft`, divided by its timesteps), measured by this run.  Returns (bytes per timestep or None, provenance dict)."""
    t0 = time.perf_counter()
    argv = ["--mode", "ft", "--steps", str(steps), "--warmup", str(warmup)] + (["--ft-groups", str(args.ft_groups)] if getattr(args, "ft_groups", None) else [])
    got, why = _counter_passes(argv, timeout_s=300)
    if got is None:
        return None, {"measured": why}
    total = sum(2.0 * v * 1024.0 * n for n, v in got["FETCH_SIZE"].values()) + sum(v * 1024.0 * n for n, v in got["WRITE_SIZE"].values())
    def kbytes(k):   # bytes of all dispatches of kernel k over the child run
        nf, f = got["FETCH_SIZE"].get(k, (0, 0.0))
        nw, w = got["WRITE_SIZE"].get(k, (0, 0.0))
        return (2.0 * f * nf + w * nw) * 1024.0
    top = sorted(((kbytes(k) / (steps + warmup), k[:60]) for k in set(got["FETCH_SIZE"]) | set(got["WRITE_SIZE"])), reverse=True)[:6]
    return total / (steps + warmup), {"measured": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of `--mode ft`, every dispatch of the run (its set-up and "
                                                  "one Adam step per call included)", "timesteps_per_pass": steps + warmup, "wall_s": round(time.perf_counter() - t0, 1),
                                      "correction": COUNTER_CORRECTION, "largest_kernels_bytes_per_timestep": {k: b for b, k in top}}


def measure_mg_traffic_live(args, steps=2, warmup=1):
    """HBM-side bytes of ALL kernels of the MatterGen-shaped sampler's step (FETCH_SIZE x2 + WRITE_SIZE, every dispatch of the child run),
    measured by this run.  Returns (bytes per step or None, provenance dict)."""
    t0 = time.perf_counter()
    got, why = _counter_passes(["--mode", "mg-sample", "--steps", str(steps), "--warmup", str(warmup), "--mg-batch", str(args.mg_batch), "--mg-chains", str(args.mg_chains)], timeout_s=300)
    if got is None:
        return None, {"measured": why}
    total = sum(2.0 * v * 1024.0 * n for n, v in got["FETCH_SIZE"].values()) + sum(v * 1024.0 * n for n, v in got["WRITE_SIZE"].values())
    return total / (steps + warmup), {"measured": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of `--mode mg-sample`, every dispatch of the run "
                                                  "(its set-up included: two graph builds)", "steps_per_pass": steps + warmup, "wall_s": round(time.perf_counter() - t0, 1),
                                      "correction": COUNTER_CORRECTION}


def _self_launch(n):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE in the environment): start the N ranks ourselves, one
    process per GPU, through torch.distributed.run on the loopback address, and pass their output through -- rank 0 prints
    the one JSON line.  Under an outer launcher (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MI_BENCH_CHILD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def tf32_class_leg(args, K, W):
    """extra.tf32_class_path: `bench.py --path tf32-class` as a child process (another library build cannot be loaded beside the product one)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--path", "tf32-class", "--steps", str(K), "--warmup", str(W), "--streams", str(args.streams),
           "--no-cpu-baseline", "--no-counters"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_child_env())
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": d["value"], "unit": d["unit"], "steps": d["steps"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                "roofline": {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches")},
                "terms_per_product": 1, "tolerance": "stated and tested in tests/test_gpu_tf32_class.py: 5e-4 of max|ref| on a benchmark-width forward (measured 1.6e-4), 5e-4 on accumulated gradients; the product path: 2e-5 (measured 7e-7 / 6e-6)",
                "note": "labelled secondary line, never the headline: north_star's tolerance is fp32"}
    except Exception as e:   # (never let the secondary figure take the headline down)
        return {"error": repr(e)}


def resolve_batch(args):
    """--scaling strong: ONE global batch of 256 crystals (north_star: "batch 256 ... >= 6x at 8 GPUs"), B / N crystals per rank, global ids in the noise
    counters as always (so the N-rank samples are the 1-rank samples).  --batch: the per-GPU batch directly -- the per-rank shapes of a strong-scaling run
    (128 / 64 / 32) measured on ONE GPU: profiles/r6_strong_shapes.json, DESIGN section 7.  --streams 0: the sampler's automatic chain count for the batch.
    Rebinds the module's per-GPU batch B; returns it."""
    global B
    b_global = 256
    B = b_global
    if args.scaling == "strong":
        if b_global % max(1, args.gpus) != 0:
            raise SystemExit(f"bench.py: --scaling strong: the global batch of {b_global} does not divide over {args.gpus} ranks")
        B = b_global // max(1, args.gpus)
    if args.batch:
        B = int(args.batch)
    args.batch_label = (f"global batch {b_global} sharded over {args.gpus} rank(s): {B} crystals per GPU (strong scaling)" if args.scaling == "strong" and not args.batch
                        else f"{B} crystals per GPU" + (" (a per-rank shape of the strong-scaling run, measured on one GPU)" if args.batch else ""))
    if args.streams == 0:   # automatic: the sampler's own choice for the batch's edge count (DiffCSPModule.sample)
        e_total = B * NATOM * NATOM
        args.streams = 4 if e_total >= 98304 else 3 if e_total >= 49152 else 2 if e_total >= 16384 else 1
    return B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ft-groups", type=int, default=None, help="--mode ft: crystal groups fine-tuned concurrently (default: automatic)")
    ap.add_argument("--streams", type=int, default=4, help="crystal groups of the batch sampled concurrently on separate HIP "
                    "streams (same samples: the noise is indexed by global ids); 0 = the sampler's automatic choice for the batch size")
    ap.add_argument("--path", choices=["split-gemm", "f32-gemm", "f32-fused", "tf32-class"], default="split-gemm",
                    help="arithmetic path: split-gemm (default) = fp32 products on the fp16 matrix pipe from two pre-split fp16 planes per operand "
                         "(three MFMA terms, f32 accumulate; a -DMI_PLANES_FP16=0 build uses three bf16 planes / six terms), fp32-class accuracy; "
                         "f32-gemm / f32-fused = f32-input MFMA with the GEMM or the register-chained edge stage; tf32-class = split-gemm's kernels in the "
                         "TF32-CLASS library build (ONE fp16 x fp16 term per plane-set product, 11-bit operands: the class the reference runs after "
                         "pipeline/mat_invent.py:127) -- a labelled secondary line with its own tolerance (tests/test_gpu_tf32_class.py), never the headline")
    ap.add_argument("--mg-batch", type=int, default=256, help="--mode mg-sample: crystals per batch")
    ap.add_argument("--mg-free-chain", action="store_true", help="--mode mg-sample: time the free-running random-init chain (emptying graph) instead of "
                    "steps that each start from the physical-density state")
    ap.add_argument("--mg-chains", type=int, default=4, help="--mode mg-sample: crystal groups sampled concurrently on separate HIP streams (default 4, the "
                    "sampler's own automatic choice at this size; bit-reproducible since the library is built without packed-fp32 instructions, DESIGN 18)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak (default): 256 crystals per GPU; strong: a global batch of 256 sharded over the ranks")
    ap.add_argument("--batch", type=int, default=0, help="crystals per GPU (default 256; 256 / N with --scaling strong)")
    ap.add_argument("--force-dist", action="store_true", help="with --gpus 1: form a world-size-1 RCCL (\"nccl\") process group and run every collective of the "
                    "multi-GPU path (barriers, max-over-ranks time, flat-gradient all-reduce) with its real arguments")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 FETCH_SIZE / WRITE_SIZE child passes that fill roofline.traffic")
    ap.add_argument("--counter-child", action="store_true", help=argparse.SUPPRESS)   # (the child of those passes: the timed chain only, no JSON)
    ap.add_argument("--mode", choices=["sample", "ft", "mg-sample", "mg-ft", "sample-default", "ft-default"], default="sample",
                    help="sample: headline metric (BASELINE configs[1]); ft: fine-tune micro-steps (configs[2]/[3]), secondary")
    args = ap.parse_args()
    if args.force_dist:
        os.environ["MI_BENCH_FORCE_DIST"] = "1"
    resolve_batch(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not os.environ.get("MI_BENCH_SHARE_GPU") and torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} starts one process per GPU over RCCL, but this node shows {torch.cuda.device_count()} GPU(s)")
        sys.exit(_self_launch(args.gpus))
    if args.mode == "ft":
        return main_ft(args)
    if args.mode == "mg-sample":
        return main_mg(args)
    if args.mode == "mg-ft":
        return main_mg_ft(args)
    if args.mode in ("sample-default", "ft-default"):
        return main_reference_defaults(args)
    tf32 = args.path == "tf32-class"
    if tf32:   # the same command on the TF32-class library build (selected before anything loads the library)
        from matinvent_amd import build as _b
        os.environ["MI_LIB_PATH"] = _b.build(verbose=False, variant="tf32")
        args.path = "split-gemm"
    K, W = args.steps, args.warmup
    assert 1 <= K <= T and 0 <= W <= T, f"steps and warmup must be <= T = {T}"

    ctx = dist_setup(args.gpus)
    world, rank, local_rank, share, dist = ctx
    dev = torch.device("cuda", local_rank)

    from matinvent_amd import _lib, build as _build
    _build.build(verbose=False)
    lib = _lib.load()
    assert (lib.mi_terms_per_product() == 1) == tf32, "bench.py: the loaded library is not the arithmetic class this line is labelled with"
    _apply_env_knobs(lib)
    from matinvent_amd.cspnet import set_gemm_mode
    set_gemm_mode("split" if args.path == "split-gemm" else "f32")
    m = build_module(dev)
    m.decoder.set_edge_mode("fused_f32" if args.path == "f32-fused" else "gemm")
    na = [NATOM] * B
    N = B * NATOM
    S = max(1, args.streams)

    class Counts:  # what DiffCSPModule.sample needs of a batch; its index tables / workspaces are cached on this object
        num_atoms = torch.tensor(na)
    cb = Counts()
    skw = dict(step_lr=STEP_LR, node_offset=rank * N, graph_offset=rank * B, streams=S)  # global ids: shard-invariant noise

    def barrier():
        dist_barrier(ctx)

    # W untimed denoising steps on a throwaway state, then exactly K timed steps of the chain that
    # starts at t = T (its Philox initial state is generated outside the timed region: inputs resident)
    import gc
    gc_early = os.environ.get("MI_BENCH_GC_EARLY", "1") not in ("", "0")
    final, _ = m.sample(cb, seed=SEED_NOISE, t_start=T, t_stop=T, **skw)
    state = (final["frac_coords"], final["lattices"], final["atom_types"])
    m._coefficients(STEP_LR)
    preheat = int(os.environ.get("MI_BENCH_PREHEAT_STEPS", "0") or 0)
    if preheat > 0:   # (experiment: untimed steps of a throwaway chain in front of everything -- does a FIRST process of a lease then measure like a later one?)
        m.sample(cb, seed=SEED_NOISE + 3, t_start=T, t_stop=max(0, T - preheat), **skw)
        torch.cuda.synchronize()
    if gc_early:
        gc.collect()   # (everything the host has to do between the warm-up and the window is done BEFORE the warm-up: a generation-2 collection of a torch
        #  process takes tens of ms, during which an idle part drops its power state -- the warm-up steps are meant to be the last thing the GPU did)
    if W > 0:
        m.sample(cb, seed=SEED_NOISE + 1, t_start=T, t_stop=T - W, **skw)
    if not gc_early:
        gc.collect()
    gc.disable()   # (as timeit does: a 20-step window is 0.1 s, a generation-2 collection of a torch process several ms of it; re-enabled right behind the window)
    try:
        barrier()
        _lib.check(lib.mi_profile_enable(m.decoder._h, 1))
        t0 = time.perf_counter()
        final, _ = m.sample(cb, seed=SEED_NOISE, init=state, t_start=T, t_stop=T - K, **skw)
        barrier()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()   # (whatever happened in the window, the secondary legs run with the collector on)
    import ctypes as C
    n_launch, tot_ms, union_ms = C.c_int64(), C.c_double(), C.c_double()
    _lib.check(lib.mi_profile_read(m.decoder._h, C.byref(n_launch), C.byref(tot_ms), C.byref(union_ms)))
    _lib.check(lib.mi_profile_enable(m.decoder._h, 0))
    if args.counter_child:   # (under rocprofv3 --pmc: the dispatches are what is wanted, nothing is printed)
        torch.cuda.synchronize()
        return
    finite = all(bool(torch.isfinite(v).all()) for v in (final["frac_coords"], final["lattices"], final["atom_types"]))
    sat = _lib.saturation_events(reset=True)   # fp16-plane conversions that clamped during the run (0 = the format held)

    elapsed = dist_max_time(ctx, elapsed)

    if rank == 0:
        value = world * B * K / (T * elapsed)
        E = (B // S) * NATOM * NATOM                               # edges one bracketed launch processes (one stream's group)
        f_exec, f_alg = edge_flops_per_edge(pairs=args.path == "split-gemm")
        avg_ms = tot_ms.value / max(1, n_launch.value)             # plain per-launch duration (what rocprofv3 --stats shows)
        # S chains run concurrently, so launches overlap: the rate the stage sustains = all its flops / the time during which
        # at least one instance was executing (= sum of durations when S = 1)
        busy_ms = union_ms.value if S > 1 else tot_ms.value
        fp32_equiv = n_launch.value * E * f_exec / (busy_ms * 1e-3) / 1e12   # TFLOP/s of fp32 multiply-adds the stage delivers
        if args.path == "split-gemm":
            # every fp32 product is issued as THREE fp16 MFMA products (two-plane fp16 operands; six bf16 products in the three-plane
            # bf16 build): price the matrix pipe with what it executes (fp16 and bf16 MFMA have the same dense peak)
            terms = lib.mi_terms_per_product() if lib.mi_plane_format() == 2 else 6
            kernel, issued, peak = "edge_gemm1b_kernel (pair mode) + edge_gemm2b_kernel (edge MLP of one layer: Fourier-block GEMM over atom pairs + second-linear GEMM over edges with the edge -> node sum)", terms * fp32_equiv, PEAK_BF16_MFMA_TFLOPS
            dtype = ("f32 via 2-plane fp16 split (3 fp16 MFMA terms, f32 accumulate)" if terms == 3 else
                     "TF32-CLASS: fp16 x fp16 leading term only (11-bit operands, f32 accumulate) -- NOT the headline arithmetic" if terms == 1 else
                     "f32 via 3-plane bf16 split (6 bf16 MFMA terms, f32 accumulate)")
        else:
            kernel = "edge_mlp_fwd_kernel<512>" if args.path == "f32-fused" else "gemm_nt_kernel<128,64> x2 (edge MLP of one layer)"
            issued, peak, dtype = fp32_equiv, PEAK_F32_MFMA_TFLOPS, "f32"
        y_eval = bytes_per_crystal_eval(NATOM)
        # HBM-side bytes per launch, MEASURED BY THIS RUN: two short child runs of this same command line under rocprofv3 (--pmc
        # FETCH_SIZE, then --pmc WRITE_SIZE: separate passes, corrected as the MI355X guide prescribes), after the timed region.  A
        # property of the code that was just timed -- no committed file can go stale behind a kernel change.  null (with the reason)
        # when rocprofv3 is missing or a pass fails; --no-counters skips the passes.
        traffic, traffic_src = None, {"measured": "skipped (--no-counters)" if args.no_counters else "not applicable to this path / world size"}
        if args.path == "split-gemm" and world == 1 and not args.no_counters:
            traffic, traffic_src = measure_traffic_live(args)
        out = {
            "metric": "crystal structures/sec (1000-step reverse diffusion)", "value": value, "unit": "structures/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: 1000-step reverse sampler, {args.batch_label} x 20 atoms, "
                                   "2 score-net evals/step (DiffCSP CSPNet H=512 L=6 F=128 fc edges; MatterGen arithmetic is "
                                   "un-vendored/parity-unpinned); a bench step = one denoising step over the batch",
                       "batch_per_gpu": B, "atoms_per_cell": NATOM, "T": T, "evals_per_step": 2, "path": args.path,
                       "concurrent_chains": S, "comm_backend": (dist.get_backend() if dist is not None else None), "world_size": world,
                       "weights": "random-init seed 0, heads x1e-2", "noise": "philox seed 1234", "final_state_finite": finite,
                       "fp16_plane_saturation_events": sat,
                       "record": False,
                       "record_note": "per-step log-probabilities not computed and the T+1 states not kept in the timed region (SURVEY 8(d) config 2: recording off "
                                      "for timing, on for parity); the reference computes and keeps both on every step (models/diffcsp/diffusion.py:353-390) -- "
                                      "extra.recording_chain is the same window with record=True",
                       "gc_disabled_in_window": True},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": issued, "peak": peak,
                         "unit": "TFLOP/s", "frac": issued / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": int(n_launch.value), "avg_launch_ms": avg_ms, "concurrent_streams": S,
                         "stage_busy_ms": busy_ms, "stage_busy_share_of_timed_region": busy_ms / (elapsed * 1e3),
                         "achieved_fp32_equivalent": fp32_equiv,
                         "peak_sustained_on_random_operands": SUSTAINED_F16_MFMA_TFLOPS if args.path == "split-gemm" else None,
                         "frac_of_sustained": issued / SUSTAINED_F16_MFMA_TFLOPS if args.path == "split-gemm" else None,
                         "flops_per_launch_executed": E * f_exec, "flops_per_launch_section8d": E * f_alg,
                         "achieved_section8d": n_launch.value * E * f_alg / (busy_ms * 1e-3) / 1e12,
                         "note": "one 'launch' = the per-edge MLP of one layer (Fourier block K=6F + 2nd linear K=H over the edges of one "
                                 "stream's crystal group); with concurrent_streams > 1 launches overlap, so achieved = flops of all "
                                 "launches / union of their event-bracketed execution intervals (avg_launch_ms is the plain per-launch "
                                 "duration). "
                                 "achieved = matrix-pipe flops actually issued; achieved_fp32_equivalent = the fp32 multiply-adds "
                                 "delivered; the section-8(d) figure also counts the h_i/h_j/gram columns that this build evaluates "
                                 "once per node instead of once per edge"},
            "end_to_end": {"tflops_section8d": 5.893e9 * 2 * B * K / elapsed / 1e12 * world,
                           "hbm_frac_section8d": (y_eval * 2 * B * K / elapsed) / (PEAK_HBM_TBPS * 1e12),
                           "edge_stage_share_of_step": busy_ms * 1e-3 / elapsed},
        }
        if not args.no_cpu_baseline and world == 1:
            if args.path == "split-gemm":
                # the precision trade, visible in every record: the same workload on the exact-fp32 path (f32-input MFMA, register-
                # chained edge stage, bit-for-bit an fp32 fma chain), a short run outside the timed region
                set_gemm_mode("f32")
                m.decoder.set_edge_mode("fused_f32")
                m.sample(cb, seed=SEED_NOISE + 2, t_start=T, t_stop=T - 2, **dict(skw, streams=1))
                torch.cuda.synchronize()
                kx = min(K, 10)
                tx = time.perf_counter()
                m.sample(cb, seed=SEED_NOISE + 2, t_start=T, t_stop=T - kx, **dict(skw, streams=1))
                torch.cuda.synchronize()
                tx = time.perf_counter() - tx
                out["extra"] = {"exact_fp32_path": {"value": B * kx / (T * tx), "unit": "structures/s", "steps": kx, "path": "f32-fused",
                                                    "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}}
                set_gemm_mode("split")
                m.decoder.set_edge_mode("gemm")
                # what the timed region leaves out, priced: the same K steps from the same state with record=True -- every step's state kept on the
                # device and its three log-probabilities computed (21-image wrapped-normal sums, block reductions), as diffusion.py:353-390 does
                try:
                    m.sample(cb, seed=SEED_NOISE + 1, t_start=T, t_stop=T - 2, record=True, **skw)
                    torch.cuda.synchronize()
                    tr = time.perf_counter()
                    fin_r, traj_r = m.sample(cb, seed=SEED_NOISE, init=state, t_start=T, t_stop=T - K, record=True, **skw)
                    torch.cuda.synchronize()
                    tr = time.perf_counter() - tr
                    same = all(bool(torch.equal(fin_r[k], final[k])) for k in ("frac_coords", "lattices", "atom_types"))
                    kept = sorted(k for k in traj_r[T - K + 1]) if (T - K + 1) in traj_r else []
                    out["extra"]["recording_chain"] = {"value": B * K / (T * tr), "unit": "structures/s", "steps": K, "ms_per_step": tr * 1e3 / K, "record": True,
                                                       "relative_to_headline": (B * K / (T * tr)) / value, "states_kept": len(traj_r), "fields_per_state": kept,
                                                       "final_state_bit_identical_to_timed_chain": same,
                                                       "note": "record=True: states of all steps retained + log_prob_l / log_prob_t / log_prob_x per step (for t > 1)"}
                    del fin_r, traj_r
                except Exception as e:   # (never let the secondary figure take the headline down)
                    out["extra"]["recording_chain"] = {"error": repr(e)}
                # the arithmetic class the REFERENCE runs after its first fine-tune step (torch.set_float32_matmul_precision("high"),
                # pipeline/mat_invent.py:127: TF32 on its hardware), as a LABELLED secondary line: the same command on the TF32-class library
                # build (one term per plane-set product; its own tolerance is stated and tested in tests/test_gpu_tf32_class.py), in a child process
                out["extra"]["tf32_class_path"] = tf32_class_leg(args, min(K, 20), W)
                # the MatterGen-LABELLED form of the same config (self-consistent, parity-unpinned vs upstream): a short run of its own
                # sampler, outside the timed region, so that every driver-run record carries it next to the pinned headline
                try:
                    mg = measure_mg(args, 6, 1, ctx)
                    out["extra"]["mattergen_shaped_sampler"] = {k: mg[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "hbm_roofline")}
                    out["extra"]["mattergen_shaped_sampler"].update(edges_first_step=mg["config"]["edges_first_step"], edges_last_step=mg["config"]["edges_last_step"],
                                                                    parity="self-consistent, PARITY-UNPINNED vs upstream", dense_layer_frac_of_mfma_peak=mg["roofline"]["frac"])
                except Exception as e:   # (never let the secondary figure take the headline down)
                    out["extra"]["mattergen_shaped_sampler"] = {"error": repr(e)}
                # BASELINE configs[2] in every driver-run record: a short fine-tune leg (B = 256, ONE whole accumulation window -- 50 timesteps
                # and the Adam step that closes it, the reference's own ratio: pipeline/mat_invent.py:150-177, configs/pipeline/mat_invent.yaml;
                # rounds 2-3 timed 20 timesteps per Adam step, 2.5 x the reference's optimizer work per timestep -- + 3 warm-up), with its own
                # roofline and CPU baseline
                try:
                    del m
                    torch.cuda.empty_cache()
                    ftl = measure_ft(args, 50, 3, ctx, cpu_budget_s=8.0)
                    out["extra"]["fine_tune"] = {k: ftl[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "cpu_baseline") if k in ftl}
                    out["extra"]["fine_tune"]["workload"] = ftl["config"]["workload"]
                    out["extra"]["fine_tune"]["adam_steps_in_timed_region"] = ftl["config"]["adam_steps_in_timed_region"]
                    # the same leg on rounds 2-3's protocol (20 timesteps per Adam step), so that the records stay comparable across rounds
                    saved = args.no_cpu_baseline, args.no_counters
                    args.no_cpu_baseline = args.no_counters = True
                    try:
                        ft20 = measure_ft(args, 20, 3, ctx)
                        out["extra"]["fine_tune"]["value_20step"] = ft20["value"]
                    finally:
                        args.no_cpu_baseline, args.no_counters = saved
                except Exception as e:
                    out["extra"]["fine_tune"] = {"error": repr(e)}
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
