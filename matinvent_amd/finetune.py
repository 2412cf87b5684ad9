"""The reward-weighted fine-tune step (MatInvent.ft_step, pipeline/mat_invent.py:125-189).

    loss = mean_b( r_b * L_dsm(b, t) + sigma * (1.1 - r_b) * ||pred_agent - pred_prior||^2 ) / accum_steps

accumulated over `accum_steps` timesteps per optimizer step, `epochs` passes over the `timesteps`
grid, a fresh Adam per call.  Differences from the reference, none of which change the update:
  * the three `.item()` host syncs per timestep (:168-170) become device-side accumulators read once
    per epoch;
  * data parallel: every rank holds a contiguous shard of the fine-tune set, scales its local sum by
    1/(B_global * accum_steps) (the reference's `.mean()` over the whole batch, :163) and one flat
    all-reduce of the gradient precedes each optimizer step; every rank applies the same fused Adam.
"""
import logging

import torch

from . import _lib
from .data import CrystalBatchData, CrystalDataset
from .dist import allreduce_flat_, rank_world, shard_range
from .optim import FusedAdam


def _sched_host(agent):
    """Host copies of the noise schedules (made once per module): indexing the device tensors costs a device sync per scalar and
    timestep, which the launch-bound small-set regime cannot afford."""
    bufs = (agent.beta_scheduler.alphas_cumprod, agent.sigma_scheduler.sigmas, agent.sigma_scheduler.sigmas_norm)
    key = tuple((b.data_ptr(), b._version) for b in bufs)  # a checkpoint load after the first ft_step changes the buffers
    h = agent.__dict__.get("_mi_sched_host")
    if h is None or h[0] != key:
        ac = bufs[0].detach().cpu()
        h = agent.__dict__["_mi_sched_host"] = (key, (torch.sqrt(ac).tolist(), torch.sqrt(1.0 - ac).tolist(),
                                                      bufs[1].detach().cpu().tolist(), bufs[2].detach().cpu().tolist()))
    return h[1]


def _fused_micro_step(agent, prior, batch, time_idx, noise, sigma, n_global, accum_steps, grad, stats, call_id=None, aux_stream=None):
    """One timestep through mi_ft_micro_step on the current stream; accumulates into `grad` (+=) and `stats` (device, 3
    floats).  `call_id` = the noise-stream call counter (one value per timestep, shared by every crystal group)."""
    import ctypes as C
    from . import _lib
    from .cspnet import _ptr, _stream
    lib = _lib.load()
    dev = agent.device
    T = agent.beta_scheduler.timesteps
    t = T - int(time_idx)
    na = batch.__dict__.setdefault("_mi_na", batch.num_atoms.cpu())  # host copy made once: no device sync per timestep
    ab, pb = agent._batch_for(na), prior._batch_for(na)
    agent.decoder.sync()
    prior.decoder.sync()
    sc0, sc1, ssig, ssn = _sched_host(agent)
    c0, c1, sig, sn = sc0[t], sc1[t], ssig[t], ssn[t]
    cache = batch.__dict__.setdefault("_mi_dev", {})
    if not cache:
        f = lambda x: x.to(dev, torch.float32).contiguous()
        cache.update(lengths=f(batch.lengths), angles=f(batch.angles), frac=f(batch.frac_coords),
                     types=batch.atom_types.to(dev, torch.int32).contiguous(), reward=f(batch.reward))
    nz = (None, None, None) if noise is None else tuple(x.to(dev, torch.float32).contiguous() for x in noise)
    if call_id is None:
        agent._noise_calls = getattr(agent, "_noise_calls", 0) + 1
        call_id = agent._noise_calls
    _lib.check(lib.mi_ft_micro_step(agent.decoder._h, ab._h, prior.decoder._h, pb._h, _ptr(cache["lengths"]), _ptr(cache["angles"]),
                                    _ptr(cache["frac"]), _ptr(cache["types"]), _ptr(cache["reward"]), _ptr(agent.time_embedding.freqs), t,
                                    c0, c1, sig, sn, getattr(agent, "noise_seed", 0), call_id & 0xFFFFFFFF, _ptr(nz[0]),
                                    _ptr(nz[1]), _ptr(nz[2]), agent.cost_lattice, agent.cost_coord, agent.cost_type, sigma, n_global,
                                    accum_steps, _ptr(grad), _ptr(stats), None, None, _stream(),
                                    C.c_void_p(aux_stream.cuda_stream) if aux_stream is not None else None), "mi_ft_micro_step")


def _stacked_micro_steps(agent, prior, batch, time_idxs, noises, sigma, n_global, accum_steps, grad, stats, aux_stream=None):
    """`len(time_idxs)` consecutive timesteps of one accumulation window as ONE micro-step over a batch holding that many replicas
    of the fine-tune set (mi_ft_micro_steps_stacked): the same gradient and statistics as calling _fused_micro_step once per
    timestep, with a fraction of the kernel launches -- a single timestep of a small set is bound by the host's launch rate."""
    import ctypes as C
    from . import _lib
    from .cspnet import _ptr, _stream
    lib = _lib.load()
    dev = agent.device
    k = len(time_idxs)
    T = agent.beta_scheduler.timesteps
    ts = [T - int(i) for i in time_idxs]
    num_atoms = batch.__dict__.setdefault("_mi_na", batch.num_atoms.cpu()).repeat(k)
    ab, pb = agent._batch_for(num_atoms), prior._batch_for(num_atoms)
    agent.decoder.sync()
    prior.decoder.sync()
    sc0, sc1, ssig, ssn = _sched_host(agent)
    arr = lambda ctype, vals: (ctype * k)(*vals)
    c0, c1 = arr(C.c_float, [sc0[t] for t in ts]), arr(C.c_float, [sc1[t] for t in ts])
    sig, sn = arr(C.c_float, [ssig[t] for t in ts]), arr(C.c_float, [ssn[t] for t in ts])
    cache = batch.__dict__.setdefault("_mi_dev_stacked", {})
    if k not in cache:
        f = lambda x: x.to(dev, torch.float32).contiguous()
        rep = lambda x: x.repeat(k, *([1] * (x.dim() - 1))).contiguous()
        cache[k] = dict(lengths=rep(f(batch.lengths)), angles=rep(f(batch.angles)), frac=rep(f(batch.frac_coords)),
                        types=rep(batch.atom_types.to(dev, torch.int32)), reward=rep(f(batch.reward)))
        while len(cache) > 3:
            cache.pop(next(iter(cache)))
    c = cache[k]
    nz = (None, None, None)
    if noises is not None:  # injected noise (parity tests): the replicas' arrays one after the other
        nz = tuple(torch.cat([n[j].to(dev, torch.float32) for n in noises]).contiguous() for j in range(3))
    call0 = getattr(agent, "_noise_calls", 0) + 1  # replica j uses the call id of its own timestep
    agent._noise_calls = call0 + k - 1
    _lib.check(lib.mi_ft_micro_steps_stacked(agent.decoder._h, ab._h, prior.decoder._h, pb._h, _ptr(c["lengths"]), _ptr(c["angles"]),
                                             _ptr(c["frac"]), _ptr(c["types"]), _ptr(c["reward"]), _ptr(agent.time_embedding.freqs), k,
                                             arr(C.c_int, ts), c0, c1, sig, sn, getattr(agent, "noise_seed", 0), call0 & 0xFFFFFFFF,
                                             _ptr(nz[0]), _ptr(nz[1]), _ptr(nz[2]), agent.cost_lattice, agent.cost_coord, agent.cost_type,
                                             sigma, n_global, accum_steps, _ptr(grad), _ptr(stats), _stream(),
                                             C.c_void_p(aux_stream.cuda_stream) if aux_stream is not None else None),
               "mi_ft_micro_steps_stacked")


MAX_STACK = 16  # MI_MAX_STACK of the C ABI
FT_PRIOR_AUX = False  # grouped path: the frozen prior's forward of every group on an auxiliary stream (measured: see DESIGN 20)
WGRAD_WINDOW = 16  # micro-steps whose node-level weight gradients are contracted together (mi_batch_set_wgrad_window); 0 = off


def auto_groups(e_total: int) -> int:
    """Concurrent crystal groups ft_step picks for a local set with `e_total` directed edges (measured at 256 x 20 atoms, round 3:
    16.1k / 16.7k / 17.3k / 13.9k / 11.7k crystal-timesteps/s with 2 / 3 / 4 / 6 / 8 groups)."""
    # (round 6, after the fused backward chain and the contractions' share of the chip: 128 crystals x 20 atoms on 1 / 2 / 3 groups 16.3k / 17.0k / 17.6k,
    #  64 crystals 12.65k / 12.46k / 12.1k, 256 crystals on 3 / 4 / 5 groups 21.3k / 21.7k / 17.0k: profiles/r6_strong_shapes.json)
    return 4 if e_total >= 90000 else 3 if e_total >= 40000 else 1


def _stack_plan(e_one, accum_steps, timesteps, stack):
    """Chunk sizes of the timestep loop: every chunk lies inside one accumulation window.  stack=None: as many timesteps per
    chunk as bring the stacked batch to ~32k edges (beyond that a micro-step is no longer launch-bound), evenly sized."""
    if stack is None:
        stack = max(1, min(MAX_STACK, 32768 // max(1, e_one)))
    stack = max(1, min(int(stack), MAX_STACK))
    plan, t = [], 0
    while t < timesteps:
        w = min(accum_steps - t % accum_steps, timesteps - t)  # timesteps left in this window
        n = -(-w // stack)                                     # chunks for them, sized evenly
        sizes = [w // n + (1 if j < w % n else 0) for j in range(n)]
        plan += sizes
        t += w
    return plan


def ft_step(agent, prior, data_list, rewards, cfg, device=None, noise_fn=None, log=logging.info, fused=True, groups=None, stack=None):
    """cfg needs: lr, accum_steps, epochs, timesteps, sigma (attribute or key access).
    `noise_fn(epoch, t)` -> (rand_l, rand_x, rand_t) injects noise (parity tests); default Philox.
    fused=True (default) enqueues each timestep through mi_ft_micro_step (noise, both forwards, the fused
    loss / penalty / gradient-seed kernel and the backward in one C call, no autograd graph); fused=False
    drives the same arithmetic through the reference's module surface (add_noise / calc_sample_loss /
    calc_kl_reg + autograd), which is what the parity tests compare it with.
    `groups` (fused path): the local fine-tune set is cut into that many contiguous crystal groups whose micro-steps are
    enqueued on separate HIP streams and run concurrently, each accumulating into its own gradient buffer (summed before
    the optimizer step) -- the same arithmetic as data-parallel ranks, inside one GPU: one group's node-level and
    reduction kernels overlap the other's large GEMMs.  None = automatic (2-3 for large sets; measured at 256 x 20 atoms:
    16.1k / 16.7k / 17.3k / 13.9k crystal-timesteps/s with 2 / 3 / 4 / 6 groups, round 3).
    `stack` (fused path, single group): up to that many consecutive timesteps of an accumulation window run as ONE stacked
    micro-step (the weights only change at the optimizer step, so they are independent; same noise, same gradient up to fp32
    summation order).  None = automatic (small sets, which are bound by the host's launch rate); 1 = off."""
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    lr, accum_steps, epochs, timesteps, sigma = get("lr"), int(get("accum_steps")), int(get("epochs")), int(get("timesteps")), get("sigma")
    device = device or agent.device
    rank, world = rank_world()
    n_global = len(data_list)
    lo, hi = shard_range(n_global, rank, world)
    if hasattr(agent, "collate"):   # MatterGen-shaped module: its own records / collate, the reference's loop over the module surface
        return _ft_step_module_surface(agent, prior, data_list, rewards, lo, hi, n_global, lr, accum_steps, epochs, timesteps, sigma, device, noise_fn,
                                       log, rank)
    dataset = CrystalDataset(data_list, rewards)
    if hi == lo:
        # fewer crystals than ranks (the fine-tune set is top-k + replay and shrinks when the validity filter keeps few samples):
        # this rank has nothing to differentiate, but must take part in every all-reduce and apply every optimizer step
        return _ft_step_empty_shard(agent, n_global, lr, accum_steps, epochs, timesteps, log, rank)
    # one batch holding the whole (local shard of the) fine-tune set (:129-133); order is irrelevant to the update
    batch = CrystalBatchData([dataset[i] for i in range(lo, hi)]).to(device)
    node_lo = sum(d.num_atoms for d in dataset.data_list[:lo])
    agent.shard_offsets = prior.shard_offsets = (node_lo, lo)
    theta = agent.decoder.theta
    if groups is None:
        groups = auto_groups(sum(d.num_atoms ** 2 for d in dataset.data_list[lo:hi])) if fused else 1
    groups = max(1, min(int(groups), hi - lo)) if fused else 1
    if groups > 1:
        return _ft_step_grouped(agent, prior, dataset, lo, hi, node_lo, n_global, groups, lr, accum_steps, epochs, timesteps, sigma, device,
                                noise_fn, log, rank)
    optimizer = FusedAdam([theta], lr=lr)  # fresh every call (:136)
    stats = []
    aux = None
    if fused:  # a single (small) group: fork the frozen prior's forward onto a second, really concurrent stream
        from .streams import concurrent_streams
        aux = concurrent_streams(2, device)[1]
    for epoch in range(epochs):
        agent.train()
        optimizer.zero_grad(set_to_none=False) if theta.grad is not None else None
        acc = torch.zeros(3, device=device)  # loss, loss_diff, loss_kl accumulators (device side)
        t = -1
        if fused:
            if theta.grad is None:
                theta.grad = torch.zeros_like(theta)
            t0 = 0
            for k in _stack_plan(sum(d.num_atoms ** 2 for d in dataset.data_list[lo:hi]), accum_steps, timesteps, stack):
                tidx = list(range(t0, t0 + k))
                noises = None if noise_fn is None else [noise_fn(epoch, i) for i in tidx]
                if k == 1:
                    _fused_micro_step(agent, prior, batch, t0, None if noises is None else noises[0], sigma, n_global, accum_steps, theta.grad,
                                      acc, aux_stream=aux)
                else:
                    _stacked_micro_steps(agent, prior, batch, tidx, noises, sigma, n_global, accum_steps, theta.grad, acc, aux_stream=aux)
                t0 += k
                t = t0 - 1
                if (t + 1) % accum_steps == 0:
                    allreduce_flat_(theta.grad)
                    optimizer.step()
                    optimizer.zero_grad(set_to_none=False)
        for t in (() if fused else range(timesteps)):
            noise = None if noise_fn is None else noise_fn(epoch, t)
            noised = agent.add_noise(batch, t, noise=noise)                       # :152
            sample_loss, agent_pred = agent.calc_sample_loss(noised)              # :153
            with torch.no_grad():
                _, prior_pred = prior.calc_sample_loss(noised)                    # :154 (frozen prior)
            adv = batch.reward
            loss_diff = adv * sample_loss                                         # :158
            kl_term = agent.calc_kl_reg(agent_pred, prior_pred, batch)            # :160
            loss_kl = kl_term * (1.1 - batch.reward)                              # :161
            loss = (loss_diff + loss_kl * sigma).sum() / (n_global * accum_steps)  # == .mean() / accum_steps (:163)
            loss.backward()
            with torch.no_grad():
                acc += torch.stack([loss.detach() * accum_steps, loss_diff.detach().sum(), loss_kl.detach().sum()])
            if (t + 1) % accum_steps == 0:                                        # :165-167
                allreduce_flat_(theta.grad)
                optimizer.step()
                optimizer.zero_grad(set_to_none=False)
        if (t + 1) % accum_steps != 0:                                            # :176-177
            allreduce_flat_(theta.grad)
            optimizer.step()
            optimizer.zero_grad(set_to_none=False)
        a = _epoch_reduce(acc, "ft_step")
        d = dict(loss=a[0] / timesteps, loss_diff=a[1] / timesteps / n_global, loss_kl=a[2] / timesteps / n_global)
        stats.append(d)
        if rank == 0:
            log(f"Epoch {epoch}: " + ", ".join(f"{k}: {v:.4f}" for k, v in d.items()))
    return stats


def _epoch_reduce(acc, where):
    """End of an epoch on every rank: sum the loss accumulators over the ranks and, in the same collective, the ranks' saturation counts
    of the two-plane fp16 format -- a rank-local check would raise on one rank while its peers go on to the next epoch's all-reduces
    (a hang instead of an error).  Returns the accumulators as a list; raises FloatingPointError on EVERY rank when any rank saturated."""
    from . import _lib
    n_local = _lib.saturation_events(reset=True)   # (synchronises the device: the epoch's kernels have finished)
    buf = torch.cat([acc.to(torch.float32), torch.tensor([float(min(n_local, 1 << 24))], device=acc.device)])
    allreduce_flat_(buf)
    a = buf.tolist()  # the only host read of the epoch
    if a[-1] > 0:
        raise FloatingPointError(
            f"{where}: {int(a[-1])} operand conversions (summed over the ranks; {n_local} on this one) saturated the two-plane fp16 format "
            "(values beyond 65504 / scale, or NaN / inf upstream): the results are outside the stated fp32-class tolerance.  Rebuild with "
            "MI_EXTRA_FLAGS=-DMI_PLANES_FP16=0 (three bf16 planes, no range limit) or use --path f32-gemm")
    return a[:-1]


PRIOR_ON_AUX_STREAM = True   # module-surface loop: the frozen prior's forward concurrently with the agent's


def _ft_step_module_surface(agent, prior, data_list, rewards, lo, hi, n_global, lr, accum_steps, epochs, timesteps, sigma, device, noise_fn, log, rank):
    """pipeline/mat_invent.py:136-189 literally, over the module surface (add_noise / calc_sample_loss / calc_kl_reg; the network is
    one differentiable op with a hand-written backward), with the fused Adam on the flat parameter vector, device-side loss
    accumulators and the data-parallel scaling / all-reduce of ft_step."""
    theta = agent.decoder.theta
    if hi == lo:
        return _ft_step_empty_shard(agent, n_global, lr, accum_steps, epochs, timesteps, log, rank)
    # The local shard as chunks of at most FT_CHUNK_ATOMS atoms: the training forward keeps every activation for the backward, and the
    # whole benchmark set (256 crystals x 20 atoms: 171 GB of activations + as much again for their gradients) does not fit one GPU.
    # The update is linear in the per-crystal losses (sum / (n_global * accum_steps)), the noise is indexed by global atom / crystal ids,
    # so chunking changes nothing but the summation order of the gradient.  Injected noise (parity tests) spans the shard: one chunk.
    from .mattergen import FT_CHUNK_ATOMS
    bounds, c0, atoms = [], lo, 0
    for i in range(lo, hi):
        n_i = int(data_list[i].num_atoms)
        if i > c0 and atoms + n_i > FT_CHUNK_ATOMS and noise_fn is None:
            bounds.append((c0, i))
            c0, atoms = i, 0
        atoms += n_i
    bounds.append((c0, hi))
    chunks = []
    for (a_, b_) in bounds:
        cb = agent.collate(data_list[a_:b_], None if rewards is None else list(rewards[a_:b_])).to(device)
        chunks.append((cb, (sum(d.num_atoms for d in data_list[:a_]), a_)))
    optimizer = FusedAdam([theta], lr=lr)
    stats = []
    aux = None
    if PRIOR_ON_AUX_STREAM and torch.device(device).type == "cuda":
        from .streams import concurrent_streams
        aux = concurrent_streams(2, device)[1]
        if aux == torch.cuda.current_stream():
            aux = concurrent_streams(2, device)[0]
    for epoch in range(epochs):
        agent.train()
        if theta.grad is not None:
            optimizer.zero_grad(set_to_none=False)
        acc = torch.zeros(3, device=device)
        t = -1
        for t in range(timesteps):
            noise = None if noise_fn is None else noise_fn(epoch, t)
            calls = getattr(agent, "_noise_calls", 0)
            for batch, offs in chunks:
                agent.shard_offsets = prior.shard_offsets = offs
                agent._noise_calls = calls                                            # (every chunk of a timestep draws from the same Philox step)
                noised = agent.add_noise(batch, t, noise=noise)                       # :152
                if aux is not None and hasattr(prior, "predict"):
                    # the frozen prior's forward on a second stream, under the agent's (separate network and batch handle).  Only the two
                    # NETWORK evaluations overlap -- kernels of this library, built without packed-fp32 instructions; the loss
                    # arithmetic is torch's own elementwise kernels, which are not, and runs after the join with nothing beside it.  (The
                    # prior's sample loss, line :154 of the reference, is computed there and never used: only its prediction is needed.)
                    aux.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(aux), torch.no_grad():
                        prior_pred = prior.predict(noised)                            # :154
                    agent_out = agent.predict(noised)
                    torch.cuda.current_stream().wait_stream(aux)
                    for v in (prior_pred.values() if isinstance(prior_pred, dict) else prior_pred):
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(torch.cuda.current_stream())
                    sample_loss, agent_pred = agent.calc_sample_loss(noised, pred=agent_out)   # :153
                else:
                    sample_loss, agent_pred = agent.calc_sample_loss(noised)          # :153
                    with torch.no_grad():
                        _, prior_pred = prior.calc_sample_loss(noised)                # :154
                loss_diff = batch.reward * sample_loss                                # :158
                loss_kl = agent.calc_kl_reg(agent_pred, prior_pred, batch) * (1.1 - batch.reward)   # :160-161
                loss = (loss_diff + loss_kl * sigma).sum() / (n_global * accum_steps)  # == .mean() / accum_steps (:163)
                loss.backward()
                with torch.no_grad():
                    acc += torch.stack([loss.detach() * accum_steps, loss_diff.detach().sum(), loss_kl.detach().sum()])
            if (t + 1) % accum_steps == 0:                                        # :165-167
                allreduce_flat_(theta.grad)
                optimizer.step()
                optimizer.zero_grad(set_to_none=False)
        if (t + 1) % accum_steps != 0:                                            # :176-177
            allreduce_flat_(theta.grad)
            optimizer.step()
            optimizer.zero_grad(set_to_none=False)
        a = _epoch_reduce(acc, "ft_step")
        d = dict(loss=a[0] / timesteps, loss_diff=a[1] / timesteps / n_global, loss_kl=a[2] / timesteps / n_global)
        stats.append(d)
        if rank == 0:
            log(f"Epoch {epoch}: " + ", ".join(f"{k}: {v:.4f}" for k, v in d.items()))
    return stats


def _ft_step_empty_shard(agent, n_global, lr, accum_steps, epochs, timesteps, log, rank):
    """ft_step of a rank whose shard is empty: zero gradient contribution, the same collectives and optimizer steps as its peers."""
    theta = agent.decoder.theta
    if theta.grad is None:
        theta.grad = torch.zeros_like(theta)
    optimizer = FusedAdam([theta], lr=lr)
    stats = []
    for epoch in range(epochs):
        theta.grad.zero_()
        # the Philox call id advances once per timestep on the ranks that hold data: keep this rank's counter in step, so that the noise of
        # later ft_steps does not depend on which ranks had empty shards before (world-size invariance of the global-id noise)
        agent._noise_calls = getattr(agent, "_noise_calls", 0) + timesteps
        n_steps = timesteps // accum_steps + (1 if timesteps % accum_steps else 0)
        for _ in range(n_steps):
            allreduce_flat_(theta.grad)
            optimizer.step()
            optimizer.zero_grad(set_to_none=False)
        acc = torch.zeros(3, device=theta.device)
        a = _epoch_reduce(acc, "ft_step")
        d = dict(loss=a[0] / timesteps, loss_diff=a[1] / timesteps / n_global, loss_kl=a[2] / timesteps / n_global)
        stats.append(d)
        if rank == 0:
            log(f"Epoch {epoch}: " + ", ".join(f"{k}: {v:.4f}" for k, v in d.items()))
    return stats


def _ft_step_grouped(agent, prior, dataset, lo, hi, node_lo, n_global, groups, lr, accum_steps, epochs, timesteps, sigma, device, noise_fn,
                     log, rank):
    """ft_step's fused path with the local set cut into `groups` crystal groups on concurrent streams (see ft_step)."""
    theta = agent.decoder.theta
    cuts = [lo + (hi - lo) * k // groups for k in range(groups + 1)]
    nodes = [node_lo]
    for k in range(groups):
        nodes.append(nodes[-1] + sum(d.num_atoms for d in dataset.data_list[cuts[k]:cuts[k + 1]]))
    batches = [CrystalBatchData([dataset[i] for i in range(cuts[k], cuts[k + 1])]).to(device) for k in range(groups)]
    offs = [(nodes[k], cuts[k]) for k in range(groups)]
    main = torch.cuda.current_stream()
    from .streams import concurrent_streams
    streams = concurrent_streams(groups, device)
    # FT_PRIOR_AUX (experiment, MI_FT_PRIOR_AUX=1): every group forks its frozen prior's forward onto a stream of its own, as the single-group path does
    aux_streams = concurrent_streams(2 * groups, device)[groups:] if FT_PRIOR_AUX else [None] * groups
    if theta.grad is None:
        theta.grad = torch.zeros_like(theta)
    grads = [theta.grad] + [torch.zeros_like(theta) for _ in range(groups - 1)]
    optimizer = FusedAdam([theta], lr=lr)  # fresh every call (:136)
    stats = []
    # The weights only change at the optimizer step, so the node-level linears' weight gradients of a run of micro-steps are ONE
    # contraction over all their rows instead of one short contraction (1.7k rows per group at 256 x 20 atoms) per micro-step: the
    # agent's batch handles keep the operand rows of up to WGRAD_WINDOW micro-steps (see include/matinvent_hip.h).
    # `cap`: the window a LATER call on these handles may ask for (its timesteps are not known here) -- the buffers are sized for it once, so that a short first
    # call (a warm-up of three timesteps) is not followed by a reallocation -- sixteen hipFree / hipMalloc pairs per group, each a device synchronisation -- at the
    # start of the next one (measured: a 20-timestep call behind a 3-timestep warm-up ran at 13.1 k instead of 19-21 k crystal-timesteps/s)
    cap = min(accum_steps, WGRAD_WINDOW) if groups <= 8 else 0   # (_batch_for caches eight handles per module)
    dec = agent.decoder
    nmax_ = max(nodes[k + 1] - nodes[k] for k in range(groups))
    slot_bytes = (7 * dec.hidden_dim * dec.num_layers + 4 * dec.hidden_dim + 203) * nmax_ * 4   # operand rows of one micro-step: node-level linears + heads / embedding
    # at most 8 GB of kept rows per group AND 24 GB over all groups (the windows of the groups are live together)
    cap = max(0, min(cap, (8 << 30) // max(1, slot_bytes), (24 << 30) // max(1, slot_bytes * groups)))
    window = min(cap, timesteps)
    handles = []
    for k in range(groups):
        agent.shard_offsets = offs[k]
        ab = agent._batch_for(batches[k].__dict__.setdefault("_mi_na", batches[k].num_atoms.cpu()))
        handles.append(ab)
    try:
        for ab in handles:
            if cap > window > 1:
                ab.set_wgrad_window(agent.decoder, cap)   # (reserves the buffers; the line below only lowers the slot count)
            ab.set_wgrad_window(agent.decoder, window if window > 1 else 0)
    except _lib.MIError as e:   # the window is an optimisation: out of memory for it -> the immediate form, not a failed fine-tune step
        if e.code != _lib.MI_ENOMEM:
            raise
        window = 0
        for ab in handles:
            ab.set_wgrad_window(agent.decoder, 0)

    def flush_wgrads():
        for k in range(groups):
            with torch.cuda.stream(streams[k]):
                handles[k].wgrad_flush(agent.decoder, grads[k])

    def optimizer_step():
        flush_wgrads()
        for k in range(groups):  # the optimizer consumes every group's gradient
            main.wait_event(streams[k].record_event())
        for g in grads[1:]:
            theta.grad.add_(g)
            g.zero_()
        allreduce_flat_(theta.grad)
        optimizer.step()
        optimizer.zero_grad(set_to_none=False)
        agent.decoder.sync()  # repack the updated weights once, on the main stream, before any group reads them
        ready = main.record_event()
        for st in streams:
            st.wait_event(ready)

    was_groups = _lib.load().mi_set_concurrent_groups(groups)   # (each group's weight-gradient contractions take their share of the chip, not all of it)
    try:
        return _ft_step_grouped_epochs(agent, prior, batches, cuts, nodes, offs, lo, node_lo, n_global, groups, accum_steps, epochs, timesteps, sigma,
                                       device, noise_fn, log, rank, theta, grads, streams, main, optimizer_step, stats, aux_streams)
    finally:
        import sys
        _lib.load().mi_set_concurrent_groups(was_groups)
        failing = sys.exc_info()[0] is not None
        try:   # (nothing pending unless an exception cut a window short -- and then its own error must not replace that exception)
            flush_wgrads()
            for ab in handles:
                ab.set_wgrad_window(agent.decoder, 0)
        except Exception:
            if not failing:
                raise


def _ft_step_grouped_epochs(agent, prior, batches, cuts, nodes, offs, lo, node_lo, n_global, groups, accum_steps, epochs, timesteps, sigma, device,
                            noise_fn, log, rank, theta, grads, streams, main, optimizer_step, stats, aux_streams=None):
    for epoch in range(epochs):
        agent.train()
        theta.grad.zero_()
        accs = [torch.zeros(3, device=device) for _ in range(groups)]
        agent.decoder.sync()
        prior.decoder.sync()
        ready = main.record_event()
        for st in streams:
            st.wait_event(ready)
        t = -1
        for t in range(timesteps):
            noise = None if noise_fn is None else noise_fn(epoch, t)
            agent._noise_calls = getattr(agent, "_noise_calls", 0) + 1
            for k in range(groups):
                nz = None
                if noise is not None:  # (rand_l [B,3,3], rand_x [N,3], rand_t [N,100]) of the local set -> this group's rows
                    g0, g1, n0, n1 = cuts[k] - lo, cuts[k + 1] - lo, nodes[k] - node_lo, nodes[k + 1] - node_lo
                    nz = (noise[0][g0:g1], noise[1][n0:n1], noise[2][n0:n1])
                agent.shard_offsets = prior.shard_offsets = offs[k]
                with torch.cuda.stream(streams[k]):
                    _fused_micro_step(agent, prior, batches[k], t, nz, sigma, n_global, accum_steps, grads[k], accs[k], call_id=agent._noise_calls,
                                      aux_stream=None if aux_streams is None else aux_streams[k])
            if (t + 1) % accum_steps == 0:
                optimizer_step()
        if (t + 1) % accum_steps != 0:
            optimizer_step()
        for k in range(groups):
            main.wait_event(streams[k].record_event())
        acc = torch.stack(accs).sum(0)
        a = _epoch_reduce(acc, "ft_step")
        d = dict(loss=a[0] / timesteps, loss_diff=a[1] / timesteps / n_global, loss_kl=a[2] / timesteps / n_global)
        stats.append(d)
        if rank == 0:
            log(f"Epoch {epoch}: " + ", ".join(f"{k}: {v:.4f}" for k, v in d.items()))
    agent.shard_offsets = prior.shard_offsets = (node_lo, lo)
    return stats
