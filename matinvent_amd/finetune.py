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

from .data import CrystalBatchData, CrystalDataset
from .dist import allreduce_flat_, rank_world, shard_range
from .optim import FusedAdam


def _fused_micro_step(agent, prior, batch, time_idx, noise, sigma, n_global, accum_steps, grad, stats):
    """One timestep through mi_ft_micro_step; accumulates into `grad` (+=) and `stats` (device, 3 floats)."""
    import ctypes as C
    from . import _lib
    from .cspnet import _ptr, _stream
    lib = _lib.load()
    dev = agent.device
    T = agent.beta_scheduler.timesteps
    t = T - int(time_idx)
    ab, pb = agent._batch_for(batch.num_atoms), prior._batch_for(batch.num_atoms)
    agent.decoder.sync()
    prior.decoder.sync()
    ac = agent.beta_scheduler.alphas_cumprod[t]
    c0, c1 = float(torch.sqrt(ac)), float(torch.sqrt(1.0 - ac))
    sig, sn = float(agent.sigma_scheduler.sigmas[t]), float(agent.sigma_scheduler.sigmas_norm[t])
    cache = batch.__dict__.setdefault("_mi_dev", {})
    if not cache:
        f = lambda x: x.to(dev, torch.float32).contiguous()
        cache.update(lengths=f(batch.lengths), angles=f(batch.angles), frac=f(batch.frac_coords),
                     types=batch.atom_types.to(dev, torch.int32).contiguous(), reward=f(batch.reward))
    nz = (None, None, None) if noise is None else tuple(x.to(dev, torch.float32).contiguous() for x in noise)
    agent._noise_calls = getattr(agent, "_noise_calls", 0) + 1
    _lib.check(lib.mi_ft_micro_step(agent.decoder._h, ab._h, prior.decoder._h, pb._h, _ptr(cache["lengths"]), _ptr(cache["angles"]),
                                    _ptr(cache["frac"]), _ptr(cache["types"]), _ptr(cache["reward"]), _ptr(agent.time_embedding.freqs), t,
                                    c0, c1, sig, sn, getattr(agent, "noise_seed", 0), agent._noise_calls & 0xFFFFFFFF, _ptr(nz[0]),
                                    _ptr(nz[1]), _ptr(nz[2]), agent.cost_lattice, agent.cost_coord, agent.cost_type, sigma, n_global,
                                    accum_steps, _ptr(grad), _ptr(stats), None, None, _stream()), "mi_ft_micro_step")


def ft_step(agent, prior, data_list, rewards, cfg, device=None, noise_fn=None, log=logging.info, fused=True):
    """cfg needs: lr, accum_steps, epochs, timesteps, sigma (attribute or key access).
    `noise_fn(epoch, t)` -> (rand_l, rand_x, rand_t) injects noise (parity tests); default Philox.
    fused=True (default) enqueues each timestep through mi_ft_micro_step (noise, both forwards, the fused
    loss / penalty / gradient-seed kernel and the backward in one C call, no autograd graph); fused=False
    drives the same arithmetic through the reference's module surface (add_noise / calc_sample_loss /
    calc_kl_reg + autograd), which is what the parity tests compare it with."""
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    lr, accum_steps, epochs, timesteps, sigma = get("lr"), int(get("accum_steps")), int(get("epochs")), int(get("timesteps")), get("sigma")
    device = device or agent.device
    rank, world = rank_world()
    n_global = len(data_list)
    lo, hi = shard_range(n_global, rank, world)
    dataset = CrystalDataset(data_list, rewards)
    # one batch holding the whole (local shard of the) fine-tune set (:129-133); order is irrelevant to the update
    batch = CrystalBatchData([dataset[i] for i in range(lo, hi)]).to(device)
    node_lo = sum(d.num_atoms for d in dataset.data_list[:lo])
    agent.shard_offsets = prior.shard_offsets = (node_lo, lo)
    theta = agent.decoder.theta
    optimizer = FusedAdam([theta], lr=lr)  # fresh every call (:136)
    stats = []
    for epoch in range(epochs):
        agent.train()
        optimizer.zero_grad(set_to_none=False) if theta.grad is not None else None
        acc = torch.zeros(3, device=device)  # loss, loss_diff, loss_kl accumulators (device side)
        t = -1
        for t in range(timesteps):
            noise = None if noise_fn is None else noise_fn(epoch, t)
            if fused:
                if theta.grad is None:
                    theta.grad = torch.zeros_like(theta)
                _fused_micro_step(agent, prior, batch, t, noise, sigma, n_global, accum_steps, theta.grad, acc)
                if (t + 1) % accum_steps == 0:
                    allreduce_flat_(theta.grad)
                    optimizer.step()
                    optimizer.zero_grad(set_to_none=False)
                continue
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
        allreduce_flat_(acc)
        a = acc.tolist()  # the only host sync of the epoch
        d = dict(loss=a[0] / timesteps, loss_diff=a[1] / timesteps / n_global, loss_kl=a[2] / timesteps / n_global)
        stats.append(d)
        if rank == 0:
            log(f"Epoch {epoch}: " + ", ".join(f"{k}: {v:.4f}" for k, v in d.items()))
    return stats
