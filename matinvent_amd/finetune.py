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


def ft_step(agent, prior, data_list, rewards, cfg, device=None, noise_fn=None, log=logging.info):
    """cfg needs: lr, accum_steps, epochs, timesteps, sigma (attribute or key access).
    `noise_fn(epoch, t)` -> (rand_l, rand_x, rand_t) injects noise (parity tests); default Philox."""
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
