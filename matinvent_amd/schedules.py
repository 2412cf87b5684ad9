"""Noise schedules of the DiffCSP diffusion (host side, built once at construction).

Mirrors models/diffcsp/scheduler.py of the reference: BetaScheduler (:54-92) and
SigmaScheduler (:95-116) keep the same registered-buffer names so checkpoints load, and
`sampler_coefficients` tabulates the per-step scalars DiffCSPModule.sample derives from them
(models/diffcsp/diffusion.py:297-343) with the same fp32 tensor ops, so the device loop needs
no host arithmetic per step.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ._lib import NCOEF


def cosine_beta_schedule(timesteps, s=0.008):
    x = torch.linspace(0, timesteps, timesteps + 1)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0.0001, 0.9999)


class BetaScheduler(nn.Module):
    def __init__(self, timesteps, scheduler_mode, beta_start=0.0001, beta_end=0.02):
        super().__init__()
        self.timesteps = timesteps
        if scheduler_mode == "cosine":
            betas = cosine_beta_schedule(timesteps)
        elif scheduler_mode == "linear":
            betas = torch.linspace(beta_start, beta_end, timesteps)
        elif scheduler_mode == "quadratic":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, timesteps) ** 2
        elif scheduler_mode == "sigmoid":
            betas = torch.sigmoid(torch.linspace(-6, 6, timesteps)) * (beta_end - beta_start) + beta_start
        else:
            raise ValueError(f"unknown scheduler_mode {scheduler_mode!r}")
        betas = torch.cat([torch.zeros([1]), betas], dim=0)
        alphas = 1.0 - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        sigmas = torch.zeros_like(betas)
        sigmas[1:] = betas[1:] * (1.0 - alphas_cumprod[:-1]) / (1.0 - alphas_cumprod[1:])
        self.register_buffer("betas", betas)
        self.register_buffer("alphas", alphas)
        self.register_buffer("alphas_cumprod", alphas_cumprod)
        self.register_buffer("sigmas", torch.sqrt(sigmas))

    def uniform_sample_t(self, batch_size, device):
        ts = np.random.choice(np.arange(1, self.timesteps + 1), batch_size)
        return torch.from_numpy(ts).to(device)


def d_log_p_wrapped_normal(x, sigma, N=10, T=1.0):
    """Score of the wrapped normal, naive 21-image sum exactly as scheduler.py:32-43."""
    num, den = 0, 0
    for i in range(-N, N + 1):
        e = torch.exp(-(x + T * i) ** 2 / 2 / sigma ** 2)
        num = num + (x + T * i) / sigma ** 2 * e
        den = den + e
    return num / den


def sigma_norm(sigma, T=1.0, sn=10000):
    """Monte-Carlo E[(d log p_wn)^2] (scheduler.py:46-51); draws from torch's global generator."""
    sigmas = sigma[None, :].repeat(sn, 1)
    x_sample = (sigma * torch.randn_like(sigmas)) % T
    return (d_log_p_wrapped_normal(x_sample, sigmas, T=T) ** 2).mean(dim=0)


class SigmaScheduler(nn.Module):
    def __init__(self, timesteps, sigma_begin=0.01, sigma_end=1.0, sigmas_norm=None):
        super().__init__()
        self.timesteps = timesteps
        self.sigma_begin = sigma_begin
        self.sigma_end = sigma_end
        sigmas = torch.FloatTensor(np.exp(np.linspace(np.log(sigma_begin), np.log(sigma_end), timesteps)))
        if sigmas_norm is None:  # like the reference: estimated at construction; checkpoints overwrite it
            sigmas_norm = torch.cat([torch.ones([1]), sigma_norm(sigmas)], dim=0)
        self.register_buffer("sigmas", torch.cat([torch.zeros([1]), sigmas], dim=0))
        self.register_buffer("sigmas_norm", torch.as_tensor(sigmas_norm, dtype=torch.float32).clone())

    def uniform_sample_t(self, batch_size, device):
        ts = np.random.choice(np.arange(1, self.timesteps + 1), batch_size)
        return torch.from_numpy(ts).to(device)


def time_embedding_freqs(dim):
    """Frequency table of SinusoidalTimeEmbeddings (diffusion.py:61-63)."""
    half_dim = dim // 2
    e = math.log(10000) / (half_dim - 1)
    return torch.exp(torch.arange(half_dim) * -e)


def fourier_freqs(n_frequencies):
    """SinusoidsEmbedding.frequencies (cspnet.py:16)."""
    return 2 * math.pi * torch.arange(n_frequencies)


def sampler_coefficients(beta: BetaScheduler, sigma: SigmaScheduler, step_lr: float) -> torch.Tensor:
    """[T+1, NCOEF] fp32 table of the scalars of diffusion.py:297-343, column order MI_C_*
    (include/matinvent_hip.h).  Row 0 is unused."""
    T = beta.timesteps
    b_al, b_ac, b_sg = beta.alphas.cpu(), beta.alphas_cumprod.cpu(), beta.sigmas.cpu()
    s_sg, s_sn = sigma.sigmas.cpu(), sigma.sigmas_norm.cpu()
    coef = torch.zeros(T + 1, NCOEF)
    for t in range(1, T + 1):
        alphas, alphas_cumprod = b_al[t], b_ac[t]
        c0 = 1.0 / torch.sqrt(alphas)
        c1 = (1 - alphas) / torch.sqrt(1 - alphas_cumprod)
        sigmas, sigma_x, sn = b_sg[t], s_sg[t], s_sn[t]
        step_corr = step_lr * (sigma_x / sigma.sigma_begin) ** 2
        std_corr = torch.sqrt(2 * step_corr)
        adj = s_sg[t - 1]
        step_pred = sigma_x ** 2 - adj ** 2
        std_pred = torch.sqrt((adj ** 2 * (sigma_x ** 2 - adj ** 2)) / (sigma_x ** 2))
        row = [c0, c1, sigmas, torch.sqrt(sn), step_corr, std_corr, step_pred, std_pred, std_corr ** 2, std_pred ** 2,
               sigmas ** 2, torch.log(sigmas)]
        coef[t, :len(row)] = torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in row])
    return coef
