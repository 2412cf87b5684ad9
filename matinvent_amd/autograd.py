"""Autograd bridge: CSPNet forward/backward through the C ABI as ONE differentiable op.

The reference's fine-tune step calls `.backward()` on a scalar built with ordinary tensor ops
on the network's three outputs (pipeline/mat_invent.py:158-164).  Keeping that surface means the
network has to be a differentiable op returning ordinary tensors: forward =
mi_cspnet_forward_train (keeps its activations inside the batch handle), backward =
mi_cspnet_backward (hand-written kernels; gradient w.r.t. the flat parameter vector only).
"""
import torch

from . import _lib
from .cspnet import MAX_ATOMIC_NUM, _ptr, _stream


class CSPNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, net, batch, t_emb, atom_types, frac, lattices):
        lib = _lib.load()
        net.sync()
        dev = theta.device
        B, N = batch.num_graphs, batch.num_nodes
        lat_out = torch.empty(B, 3, 3, device=dev)
        coord_out = torch.empty(N, 3, device=dev)
        type_out = torch.empty(N, MAX_ATOMIC_NUM, device=dev)
        _lib.check(lib.mi_cspnet_forward_train(net._h, batch._h, _ptr(t_emb), _ptr(atom_types), _ptr(frac), _ptr(lattices),
                                               _ptr(lat_out), _ptr(coord_out), _ptr(type_out), _stream()), "mi_cspnet_forward_train")
        ctx.net, ctx.batch = net, batch
        ctx.nparams = theta.numel()
        return lat_out, coord_out, type_out

    @staticmethod
    def backward(ctx, d_lat, d_coord, d_type):
        lib = _lib.load()
        net, batch = ctx.net, ctx.batch
        dev = net.theta.device
        B, N = batch.num_graphs, batch.num_nodes
        z = lambda g, *s: (torch.zeros(*s, device=dev) if g is None else g.contiguous().float())
        d_lat, d_coord, d_type = z(d_lat, B, 3, 3), z(d_coord, N, 3), z(d_type, N, MAX_ATOMIC_NUM)
        grad = torch.zeros(ctx.nparams, device=dev)
        _lib.check(lib.mi_cspnet_backward(net._h, batch._h, _ptr(d_lat), _ptr(d_coord), _ptr(d_type), _ptr(grad), _stream()),
                   "mi_cspnet_backward")
        return grad, None, None, None, None, None, None
