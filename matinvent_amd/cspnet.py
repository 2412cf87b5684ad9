"""CSPNet score network on the HIP path (host wrapper over mi_net / mi_batch).

Mirrors models/diffcsp/cspnet.py:94-294 of the reference as built by DiffCSPModule
(smooth=True, pred_type=True, fc edges).  All parameters live in ONE flat fp32 vector
(`theta`, the reference's `decoder.*` state_dict order) so the optimizer and the gradient
all-reduce touch a single buffer; `state_dict()` still exposes the reference's tensor names.
"""
import ctypes as C
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .schedules import fourier_freqs

MAX_ATOMIC_NUM = 100


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.int32), (t.device, t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_gemm_mode(mode: str):
    """Process-wide matrix path of every node-/edge-level GEMM: 'split' (default; three bf16 planes,
    six product terms on the bf16 matrix pipe, fp32-class accuracy) or 'f32' (f32-input MFMA)."""
    _lib.check(_lib.load().mi_set_gemm_mode({"f32": 0, "split": 1}[mode]), "mi_set_gemm_mode")


def set_edge_pairs(on: bool):
    """fc edge style, plane-GEMM edge stage: evaluate the Fourier block once per unordered node pair (default) or once per
    directed edge."""
    _lib.check(_lib.load().mi_set_edge_pairs(int(bool(on))), "mi_set_edge_pairs")


class CrystalBatch:
    """Index tables + workspace of one batch of crystals (mi_batch).  Replaces the PyG Batch
    bookkeeping (`num_atoms`, `batch`) and the per-call edge enumeration of gen_edges."""

    def __init__(self, net: "CSPNet", num_atoms, node_offset: int = 0, graph_offset: int = 0, edge_style: str = "fc",
                 max_neighbors: int = 20, edge_cap_per_node: int = 48):
        lib = _lib.load()
        na = [int(x) for x in (num_atoms.tolist() if torch.is_tensor(num_atoms) else num_atoms)]
        self.num_atoms_list = na
        self.num_graphs = len(na)
        self.num_nodes = sum(na)
        arr = (C.c_int * max(len(na), 1))(*na)
        h = C.c_void_p()
        self.edge_style = edge_style
        if edge_style == "knn":
            _lib.check(lib.mi_batch_create_knn(net._h, arr, len(na), node_offset, graph_offset, max_neighbors, edge_cap_per_node, C.byref(h)),
                       "mi_batch_create_knn")
        else:
            _lib.check(lib.mi_batch_create(net._h, arr, len(na), node_offset, graph_offset, C.byref(h)), "mi_batch_create")
        self._h = h
        self._dev = net.theta.device
        self._lib = lib
        dev = net.theta.device
        self.num_atoms = torch.tensor(na, dtype=torch.long, device=dev)
        self.batch = torch.repeat_interleave(torch.arange(len(na), device=dev), self.num_atoms)
        self.num_edges = int(lib.mi_batch_num_edges(h))

    # ---- knn edge style (gen_edges knn branch, cspnet.py:243-257) ----
    def build_graph(self, frac_coords, lattices) -> int:
        """Rebuild the periodic neighbour list for these coordinates (the forward does this itself); returns E''."""
        f = lambda x: x.detach().to(self._dev, torch.float32).contiguous()
        n = C.c_int64()
        fr, lat = f(frac_coords), f(lattices)  # keep both alive across the call
        _lib.check(self._lib.mi_knn_graph(self._h, _ptr(fr), _ptr(lat), _stream(), C.byref(n)), "mi_knn_graph")
        self.num_edges = int(n.value)
        return self.num_edges

    def edges(self, order: str = "reference"):
        """(edges [2,E''] int64, frac_diff [E'',3]) of the current list, as gen_edges returns them."""
        E = int(self._lib.mi_batch_num_edges(self._h))
        ei = torch.empty(2, E, dtype=torch.int32, device=self._dev)
        ev = torch.empty(E, 3, dtype=torch.float32, device=self._dev)
        _lib.check(self._lib.mi_knn_graph_read(self._h, _ptr(ei), _ptr(ev), {"reference": 0, "csr": 1}[order], _stream()), "mi_knn_graph_read")
        return ei.long(), ev

    # ---- node-level weight gradients over a window of micro-steps (mi_batch_set_wgrad_window) ----
    def set_wgrad_window(self, net: "CSPNet", micro_steps: int):
        """Keep the node-level linears' operand rows of `micro_steps` backward passes and contract them together (0: each backward
        contracts its own).  `wgrad_flush` must run before the gradient buffer is read."""
        if getattr(self, "_wgrad_window", 0) != int(micro_steps):
            _lib.check(self._lib.mi_batch_set_wgrad_window(net._h, self._h, int(micro_steps)), "mi_batch_set_wgrad_window")
            self._wgrad_window = int(micro_steps)

    def wgrad_flush(self, net: "CSPNet", grad):
        """grad (flat, the network's theta layout) += the pending micro-steps' node-level weight gradients, on the current stream."""
        _lib.check(self._lib.mi_cspnet_wgrad_flush(net._h, self._h, _ptr(grad), _stream()), "mi_cspnet_wgrad_flush")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and getattr(self, "_lib", None) is not None:
            self._lib.mi_batch_destroy(h)


class CSPNet(nn.Module):
    def __init__(self, hidden_dim=128, latent_dim=256, num_layers=4, max_atoms=100, act_fn="silu", dis_emb="sin",
                 num_freqs=10, edge_style="fc", cutoff=6.0, max_neighbors=20, ln=False, ip=True, smooth=False,
                 pred_type=False, pred_scalar=False, device=None):
        super().__init__()
        if not (smooth and pred_type) or pred_scalar or act_fn != "silu" or dis_emb != "sin" or not ip or max_atoms != MAX_ATOMIC_NUM:
            raise NotImplementedError("HIP CSPNet implements the DiffCSPModule configuration: smooth=True, pred_type=True, "
                                      "act_fn='silu', dis_emb='sin', ip=True, max_atoms=100")
        if edge_style not in ("fc", "knn"):
            raise NotImplementedError(f"edge_style={edge_style!r}")
        # `cutoff` is accepted and, as in the reference, has no effect: radius_graph_pbc overwrites it with the smallest
        # inter-plane spacing + 0.01 (utils.py:463-471)
        self.edge_style, self.cutoff, self.max_neighbors = edge_style, cutoff, max_neighbors
        self.edge_cap_per_node = 48   # knn edge style: kept neighbours per centre atom the batch handles are sized for (exceeding it is an error, never a truncation)
        self.hidden_dim, self.latent_dim, self.num_layers, self.num_freqs, self.ln = hidden_dim, latent_dim, num_layers, num_freqs, ln
        lib = _lib.load()
        self._lib = lib
        cfg = _lib.NetConfig(hidden_dim, num_layers, num_freqs, latent_dim, int(bool(ln)))
        h = C.c_void_p()
        _lib.check(lib.mi_net_create(C.byref(cfg), C.byref(h)), "mi_net_create")
        self._h = h
        self.layout = OrderedDict()
        for i in range(lib.mi_net_num_tensors(h)):
            name, off, numel, rows, cols = C.c_char_p(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
            _lib.check(lib.mi_net_param_info(h, i, C.byref(name), C.byref(off), C.byref(numel), C.byref(rows), C.byref(cols)))
            shape = (cols.value,) if (rows.value == 1 and name.value.decode().endswith(("bias", "layer_norm.weight"))) else (rows.value, cols.value)
            self.layout[name.value.decode()] = (off.value, numel.value, shape)
        n = int(lib.mi_net_num_params(h))
        device = torch.device(device if device is not None else "cuda")
        self.theta = nn.Parameter(torch.zeros(n, dtype=torch.float32, device=device))
        self.theta._mi_owner = self
        self._packed_version = -1
        self._freqs = fourier_freqs(num_freqs).float().contiguous()
        self._dirty = True
        self.reset_parameters()

    # ---- parameters -------------------------------------------------------------------------
    def views(self):
        """name -> view into theta (reference tensor names without the 'decoder.' prefix)."""
        return OrderedDict((k, self.theta.data[o:o + n].view(shape)) for k, (o, n, shape) in self.layout.items())

    @torch.no_grad()
    def reset_parameters(self):
        """nn.Linear / nn.LayerNorm default init, drawn in the reference's construction order."""
        v = self.views()
        for name, w in v.items():
            if name.endswith("layer_norm.weight"):
                w.fill_(1.0)
            elif name.endswith("layer_norm.bias"):
                w.zero_()
            elif name.endswith(".weight"):
                lin = torch.empty(w.shape)
                nn.init.kaiming_uniform_(lin, a=math.sqrt(5))
                w.copy_(lin)
                bname = name[:-len("weight")] + "bias"
                if bname in v:
                    bound = 1 / math.sqrt(w.shape[1])
                    v[bname].copy_(torch.empty(v[bname].shape).uniform_(-bound, bound))
        self._dirty = True

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        for k, w in self.views().items():
            out[prefix + k] = w.detach().clone()
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        v = self.views()
        with torch.no_grad():
            for k, w in v.items():
                key = prefix + k
                if key in state_dict:
                    w.copy_(state_dict[key].reshape(w.shape))
                elif strict:
                    missing_keys.append(key)
        self._dirty = True

    def mark_dirty(self):
        """Call after theta changed in place (optimizer step): packed weights are rebuilt lazily."""
        self._dirty = True

    def sync(self):
        """(Re)build the packed weight copies if theta changed: explicitly marked dirty (FusedAdam,
        load_state_dict) or bumped in place by a torch optimizer (tensor version counter)."""
        if self._dirty or self._packed_version != self.theta._version:
            fr = self._freqs.numpy()
            _lib.check(self._lib.mi_net_set_params(self._h, _ptr(self.theta.data), fr.ctypes.data_as(C.POINTER(C.c_float)), _stream()),
                       "mi_net_set_params")
            self._dirty = False
            self._packed_version = self.theta._version

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.theta._mi_owner = self
        self._dirty = True
        return r

    # ---- arithmetic path ------------------------------------------------------------------------
    def set_edge_mode(self, mode: str):
        """'gemm' (default): per-edge MLP as two tiled GEMMs; 'fused_f32': register-chained f32-MFMA kernel."""
        code = {"fused_f32": 0, "gemm": 1}[mode]
        _lib.check(self._lib.mi_net_set_edge_mode(self._h, code), "mi_net_set_edge_mode")
        self.edge_mode = mode

    # ---- forward ------------------------------------------------------------------------------
    def make_batch(self, num_atoms, node_offset=0, graph_offset=0) -> CrystalBatch:
        return CrystalBatch(self, num_atoms, node_offset, graph_offset, edge_style=self.edge_style, max_neighbors=self.max_neighbors,
                            edge_cap_per_node=self.edge_cap_per_node)

    def forward(self, t, atom_types, frac_coords, lattices, num_atoms, node2graph=None, batch: CrystalBatch = None):
        """Same positional signature as the reference CSPNet.forward (cspnet.py:260); `batch`
        carries the prebuilt index tables (built from num_atoms when absent).  With grad enabled
        and trainable parameters this is a differentiable op (matinvent_amd.autograd) whose
        backward produces dLoss/dtheta; otherwise the inference entry point runs."""
        if batch is None:
            batch = self.make_batch(num_atoms)
        B, N = batch.num_graphs, batch.num_nodes
        dev = self.theta.device
        f = lambda x: x.detach().to(dev, torch.float32).contiguous()
        t, atom_types, frac_coords, lattices = f(t), f(atom_types), f(frac_coords), f(lattices)
        assert t.shape == (B, self.latent_dim) and atom_types.shape == (N, MAX_ATOMIC_NUM) and frac_coords.shape == (N, 3)
        if torch.is_grad_enabled() and self.theta.requires_grad:
            from .autograd import CSPNetFunction
            return CSPNetFunction.apply(self.theta, self, batch, t, atom_types, frac_coords, lattices)
        self.sync()
        lat_out = torch.empty(B, 3, 3, device=dev)
        coord_out = torch.empty(N, 3, device=dev)
        type_out = torch.empty(N, MAX_ATOMIC_NUM, device=dev)
        _lib.check(self._lib.mi_cspnet_forward(self._h, batch._h, _ptr(t), _ptr(atom_types), _ptr(frac_coords), _ptr(lattices),
                                               _ptr(lat_out), _ptr(coord_out), _ptr(type_out), _stream()), "mi_cspnet_forward")
        return lat_out, coord_out, type_out

    def tap(self, batch: CrystalBatch, layer: int):
        out = torch.empty(batch.num_nodes, self.hidden_dim, device=self.theta.device)
        _lib.check(self._lib.mi_cspnet_tap(self._h, batch._h, layer, _ptr(out), _stream()), "mi_cspnet_tap")
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and getattr(self, "_lib", None) is not None:
            self._lib.mi_net_destroy(h)
