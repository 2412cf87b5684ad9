"""Fused Adam on the flat parameter vector (mi_adam_step).

Same update as `torch.optim.Adam(params, lr)` with its defaults, which is what
MatInvent.ft_step constructs (pipeline/mat_invent.py:136): beta=(0.9, 0.999), eps=1e-8, no
weight decay, no amsgrad, bias-corrected; state starts at zero.  One kernel over one buffer.
"""
import torch

from . import _lib
from .cspnet import _ptr, _stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._lib = _lib.load()

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                _lib.check(self._lib.mi_adam_step(_ptr(p.data), _ptr(p.grad), _ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"]), p.numel(),
                                                  st["step"], group["lr"], b1, b2, group["eps"], grad_scale, _stream()), "mi_adam_step")
                owner = getattr(p, "_mi_owner", None)
                if owner is not None:
                    owner.mark_dirty()  # packed weight copies are stale now
