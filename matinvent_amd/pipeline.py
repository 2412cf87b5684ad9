"""RL pipelines with the reference's class names, constructor arguments and method surface
(pipeline/base.py:27-136 ReinL, pipeline/mat_invent.py:17-290 MatInvent, pipeline/baseline.py Baseline).
Sampling and fine-tuning run on the HIP path; scoring, filtering and memory are the caller's plug-ins
(rewards / filters are out of scope and optional here)."""
import logging
import os
import time

import numpy as np
import torch

from . import config as C
from .dist import broadcast_object, rank_world
from .filters import invalid_filter
from .finetune import ft_step as _ft_step
from .memory import ReplayBuffer
from .structure import write_extxyz
from .suite import get_device


class ReinL:
    def __init__(self, rl_epoch, model_suite, reward, sample_cfg, finetune_cfg, save_dir, save_freq, device=None, logger=None,
                 replay=False, replay_args=None, **kwargs):
        self.rl_epoch, self.model_suite, self.reward = rl_epoch, model_suite, reward
        self.save_dir, self.save_freq, self.logger = save_dir, save_freq, logger
        self.device = get_device(device)
        self.cfg = C.create(kwargs)
        self.step, self.cost = 0, 0
        self.sample_cfg = C.merge(model_suite.sample_cfg, sample_cfg)          # pipeline wins (base.py:53-59)
        self.finetune_cfg = C.merge(model_suite.finetune_cfg, finetune_cfg)
        self.sampler = model_suite.get_sampler()
        self.models_dir = os.path.join(save_dir, "models")
        self.sample_dir = os.path.join(save_dir, "samples")
        os.makedirs(self.models_dir, exist_ok=True)
        os.makedirs(self.sample_dir, exist_ok=True)
        self.replay = ReplayBuffer(**(replay_args or {})) if replay else None

    def reward_step(self, sample_data, sample_struc, xyz_path=None, label="tmp"):
        """base.py:98-127: score, drop failed samples."""
        rewards, prop_dict, failed = self.reward.scoring((sample_struc, xyz_path), label)
        self.cost += len(sample_struc)
        ok = ~np.asarray(failed)
        data = [d for d, k in zip(sample_data, ok) if k]
        strucs = [s for s, k in zip(sample_struc, ok) if k]
        return data, strucs, np.asarray(rewards)[ok].astype(float), {k: np.asarray(v)[ok] for k, v in prop_dict.items()}


class MatInvent(ReinL):
    def __init__(self, rl_epoch, model_suite, reward, sample_cfg, finetune_cfg, topk_ratio, save_dir, save_freq=50, device=None,
                 logger=None, replay=False, replay_args=None, div_filter=False, df_args=None, **kwargs):
        super().__init__(rl_epoch=rl_epoch, model_suite=model_suite, reward=reward, sample_cfg=sample_cfg, finetune_cfg=finetune_cfg,
                         save_dir=save_dir, save_freq=save_freq, device=device, logger=logger, replay=replay, replay_args=replay_args,
                         **kwargs)
        assert 0.0 < topk_ratio <= 1.0
        self.topk_ratio = topk_ratio
        self.div_filter = div_filter  # the LTM diversity filter is out of scope; ignored
        self.load_model()

    def load_model(self):
        """mat_invent.py:62-72: agent (trainable) + frozen prior, two separate loads."""
        self.agent = self.model_suite.load_model()
        self.prior = self.model_suite.load_model()
        for p in self.agent.parameters():
            p.requires_grad = True
        for p in self.prior.parameters():
            p.requires_grad = False
        self.agent.to(self.device)
        self.prior.to(self.device)

    def sample_step(self):
        """mat_invent.py:74-123: sample, geometric validity pre-filter (device-side quantities), save the valid set as
        extxyz, optional filter callable, max_num.  MLIP relaxation / SUN metrics are out of scope."""
        rank, world = rank_world()
        kw = {k: v for k, v in self.sample_cfg.items() if k not in ("filter", "mlip_opt", "geometric_filter")}
        data, strucs = self.sampler.generate(model=self.agent, rank=rank, world_size=world, **kw)
        if self.sample_cfg.get("geometric_filter", True):  # the reference always filters (mat_invent.py:78-79)
            n_all = len(data)
            data, strucs = invalid_filter(data, strucs)
            logging.info(f"geometric pre-filter kept {len(data)} of {n_all} samples")
        if rank == 0 and getattr(self, "sample_dir", None):
            write_extxyz(strucs, os.path.join(self.sample_dir, f"step_{self.step:0>4d}_valid.extxyz"))
        flt = self.sample_cfg.get("filter")
        metrics = {}
        if callable(flt):
            data, strucs, metrics = flt(data, strucs, None)
        max_num = self.sample_cfg.get("max_num")
        if max_num and len(strucs) > max_num:
            data, strucs = data[:max_num], strucs[:max_num]
        return data, strucs, None, metrics

    def ft_step(self, data_list, rewards, baseline=None):
        return _ft_step(self.agent, self.prior, data_list, rewards, self.finetune_cfg, device=self.device)

    def rl_step(self):
        t0 = time.time()
        rank, world = rank_world()
        logging.info(f"*****   LOOP {self.step} START   *****")
        data, strucs, xyz, metrics = self.sample_step()
        if len(data) == 0:  # (the reference would fail inside reward scoring; identical on every rank, so no rank diverges)
            logging.warning("no sample passed the validity pre-filter; skipping scoring and fine-tuning for this loop")
            return
        if rank == 0:  # scoring / ranking / replay are rank-0 CPU bookkeeping; the chosen set is broadcast
            data, strucs, rewards, props = self.reward_step(data, strucs, xyz, f"step_{self.step:0>4d}")
            log = {f"{k} mean": v.mean() for k, v in props.items()}
            log.update({"reward mean": rewards.mean(), "reward std": rewards.std(), "cost": self.cost}, **metrics)
            if self.logger is not None:
                self.logger.log(log, step=self.step)
            order = np.argsort(rewards)[::-1]
            topk = order[: int(self.finetune_cfg.batch_size * self.topk_ratio)]
            ft_data, ft_reward = [data[i] for i in topk], rewards[topk]
            if self.replay is not None:
                rd, rr = self.replay.sample()
                self.replay.extend(ft_data, None, ft_reward)
                ft_data, ft_reward = ft_data + rd, np.concatenate((ft_reward, rr))
            payload = (ft_data, ft_reward)
        else:
            payload = None
        ft_data, ft_reward = broadcast_object(payload, src=0)
        self.ft_step(ft_data, ft_reward)
        logging.info(f"*****   LOOP {self.step} FINISH   *****  {(time.time() - t0) / 60:.2f} min")

    def run_rl(self):
        rank, _ = rank_world()
        for step in range(self.rl_epoch):
            self.step = step
            self.rl_step()
            if (step + 1) % self.save_freq == 0 and rank == 0:
                self.model_suite.save_model(self.agent, os.path.join(self.models_dir, f"loop_{step:0>4d}"))
        if rank == 0:
            self.model_suite.save_model(self.agent, os.path.join(self.models_dir, "final"))


class Baseline(ReinL):
    """pipeline/baseline.py: sample + score only (no fine-tuning); with the reference's 2-of-3 unpack
    bug (:78) fixed."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.agent = self.model_suite.load_model().to(self.device)

    def rl_step(self):
        kw = {k: v for k, v in self.sample_cfg.items() if k not in ("filter", "mlip_opt", "geometric_filter")}
        data, strucs = self.sampler.generate(model=self.agent, **kw)
        data, strucs, rewards, props = self.reward_step(data, strucs, None, f"step_{self.step:0>4d}")
        if self.logger is not None:
            self.logger.log({"reward mean": rewards.mean(), "reward std": rewards.std()}, step=self.step)

    def run_rl(self):
        for step in range(self.rl_epoch):
            self.step = step
            self.rl_step()
