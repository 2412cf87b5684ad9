"""ctypes binding of include/matinvent_hip.h.  No fallback: if the HIP library is missing or
a call fails, this raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_LIB_PATH") or os.path.join(HERE, "lib", "libmatinvent_hip.so")   # (MI_LIB_PATH: an A/B build of scripts/build_variant.py)

NCOEF = 16
NUM_TYPES = 100


class NetConfig(C.Structure):
    _fields_ = [("hidden_dim", C.c_int), ("num_layers", C.c_int), ("num_freqs", C.c_int), ("time_dim", C.c_int),
                ("ln", C.c_int)]


class SamplerNoise(C.Structure):
    _fields_ = [("corr_x", C.c_void_p), ("pred_l", C.c_void_p), ("pred_t", C.c_void_p), ("pred_x", C.c_void_p)]


class SamplerRecord(C.Structure):
    _fields_ = [("atom_types", C.c_void_p), ("frac_coords", C.c_void_p), ("lattices", C.c_void_p),
                ("frac_coords_mid", C.c_void_p), ("log_prob_l", C.c_void_p), ("log_prob_t", C.c_void_p),
                ("log_prob_x", C.c_void_p)]


class GemNetConfig(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("emb_atom", "emb_edge", "emb_trip", "emb_rbf", "emb_cbf", "emb_bil", "num_radial", "num_spherical",
                                       "num_blocks", "num_before_skip", "num_after_skip", "num_concat", "num_atom", "max_neighbors",
                                       "max_images")] + [("cutoff", C.c_float)]


class MGCorruption(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("sigma_min", "sigma_max", "beta_min", "beta_max", "limit_density", "limit_var_scale")] + [("d3pm_steps", C.c_int)]


class MGSamplerNoise(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("corr_pos", "corr_cell", "pred_pos", "pred_cell", "pred_u1", "pred_u2")]


_P, _I, _L, _U64, _U32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32

# name -> (restype, argtypes); mirrors include/matinvent_hip.h one to one
SIGNATURES = {
    "mi_last_error": (C.c_char_p, []),
    "mi_version": (_I, []),
    "mi_trace_push": (_I, [C.c_char_p]),
    "mi_trace_pop": (_I, []),
    "mi_net_create": (_I, [C.POINTER(NetConfig), C.POINTER(_P)]),
    "mi_net_destroy": (None, [_P]),
    "mi_net_num_params": (_L, [_P]),
    "mi_net_num_tensors": (_I, [_P]),
    "mi_net_param_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_L), C.POINTER(_L), C.POINTER(_I), C.POINTER(_I)]),
    "mi_net_set_params": (_I, [_P, _P, C.POINTER(C.c_float), _P]),
    "mi_batch_create": (_I, [_P, C.POINTER(_I), _I, _L, _L, C.POINTER(_P)]),
    "mi_batch_create_knn": (_I, [_P, C.POINTER(_I), _I, _L, _L, _I, _I, C.POINTER(_P)]),
    "mi_knn_graph": (_I, [_P, _P, _P, _P, C.POINTER(_L)]),
    "mi_knn_graph_read": (_I, [_P, _P, _P, _I, _P]),
    "mi_structure_check": (_I, [_P, _P, _P, _P, _P]),
    "mi_structure_check_offsets": (_I, [_P, _I, _P, _P, _P, _P]),
    "mi_debug_spin": (_I, [C.c_longlong, _P]),
    "mi_set_edge_pairs": (_I, [_I]),
    "mi_set_concurrent_groups": (_I, [_I]),
    "mi_knn_graph_status": (_I, [_P, _P]),
    "mi_plane_format": (_I, []),
    "mi_terms_per_product": (_I, []),
    "mi_debug_mfma_flops": (_I, [C.POINTER(C.c_double), C.POINTER(C.c_double), _I]),
    "mi_debug_fourier_pairs": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "mi_debug_set_db_min_tiles": (_I, [_I]),
    "mi_debug_set_node_planes_min_rows": (_I, [_I]),
    "mi_debug_set_tn128": (_I, [_I]),
    "mi_debug_set_tn_split_min_rows": (_I, [_I]),
    "mi_debug_set_tn_target_tiles": (_I, [_I]),
    "mi_debug_set_pair_wide": (_I, [_I]),
    "mi_debug_set_planes_small_tiles": (_I, [_I]),
    "mi_debug_set_planes_big": (_I, [_I, _I]),
    "mi_debug_set_planes_rt": (_I, [_I, _I]),
    "mi_debug_set_planes_latency": (_I, [_I]),
    "mi_debug_set_planes_dma": (_I, [_I]),
    "mi_debug_set_planes_big_seg": (_I, [_I]),
    "mi_debug_set_node_priority": (_I, [_I]),
    "mi_debug_set_node_fused": (_I, [_I]),
    "mi_debug_node_chain_clock": (_I, [_P]),
    "mi_debug_set_edge2_fused": (_I, [_I]),
    "mi_debug_set_node_train": (_I, [_I]),
    "mi_debug_set_node_bwd": (_I, [_I, _I]),
    "mi_debug_set_knn_nosync": (_I, [_I]),
    "mi_debug_set_node_split": (_I, [_I]),
    "mi_debug_set_node_cols": (_I, [_I]),
    "mi_debug_set_node_touch": (_I, [_I]),
    "mi_debug_set_heads_rows16": (_I, [_I]),
    "mi_debug_set_eval_reuse": (_I, [_I]),
    "mi_debug_set_skip": (_I, [_I]),
    "mi_debug_rt_clock": (_I, [_P, _I, _I]),
    "mi_debug_set_rt_lean": (_I, [_I]),
    "mi_debug_set_edge_fused": (_I, [_I]),
    "mi_debug_edge_fused_clock": (_I, [_P]),
    "mi_debug_edge2_clock": (_I, [_P]),
    "mi_debug_set_edge1_fused": (_I, [_I]),
    "mi_debug_edge1_clock": (_I, [_P]),
    "mi_batch_destroy": (None, [_P]),
    "mi_batch_num_nodes": (_I, [_P]),
    "mi_batch_num_edges": (_L, [_P]),
    "mi_batch_node2graph": (_P, [_P]),
    "mi_cspnet_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi_cspnet_tap": (_I, [_P, _P, _I, _P, _P]),
    "mi_time_embedding": (_I, [_P, _P, _I, _I, _P, _P]),
    "mi_sampler_init_state": (_I, [_P, _U64, _I, _P, _P, _P, _P]),
    "mi_sampler_run": (_I, [_P, _P, C.POINTER(C.c_float), _I, _I, _I, _P, _U64, C.POINTER(SamplerNoise),
                            C.POINTER(SamplerRecord), _P, _P, _P, _P]),
    "mi_philox_fill": (_I, [_U64, _U32, _U32, _L, _L, _I, _P, _P]),
    "mi_cspnet_forward_train": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi_cspnet_backward": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "mi_batch_set_wgrad_window": (_I, [_P, _P, _I]),
    "mi_cspnet_wgrad_flush": (_I, [_P, _P, _P, _P]),
    "mi_batch_wgrad_pending": (_I, [_P]),
    "mi_adam_step": (_I, [_P, _P, _P, _P, _L, _I, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "mi_add_noise": (_I, [_P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _U64, _U32, _P, _P, _P, _P, _P, _P, _P,
                          _P, _P, _P]),
    "mi_add_noise_per_crystal": (_I, [_P, _P, _P, _P, _P, _P, _U64, _U32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi_sampler_set_keep": (_I, [_P, _I, _I]),
    "mi_ft_micro_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, C.c_float, C.c_float, C.c_float, C.c_float, _U64, _U32, _P, _P, _P,
                              C.c_float, C.c_float, C.c_float, C.c_float, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mi_ft_micro_steps_stacked": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _U64, _U32, _P, _P, _P,
                                       C.c_float, C.c_float, C.c_float, C.c_float, _I, _I, _P, _P, _P, _P]),
    "mi_gemnet_create": (_I, [C.POINTER(GemNetConfig), C.POINTER(_P)]),
    "mi_gemnet_destroy": (None, [_P]),
    "mi_gemnet_num_params": (_L, [_P]),
    "mi_gemnet_num_tensors": (_I, [_P]),
    "mi_gemnet_param_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_L), C.POINTER(_L), C.POINTER(_I), C.POINTER(_I)]),
    "mi_gemnet_set_params": (_I, [_P, _P, _P]),
    "mi_gbatch_create": (_I, [_P, C.POINTER(_I), _I, _L, _L, C.POINTER(_P)]),
    "mi_gbatch_destroy": (None, [_P]),
    "mi_gbatch_set_offsets": (_I, [_P, _L, _L]),
    "mi_gemnet_graph": (_I, [_P, _P, _P, _P, _P, C.POINTER(_L)]),
    "mi_gemnet_graph_read": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi_gemnet_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mi_gemnet_backward": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "mi_debug_set_mg_f16": (_I, [_I]),
    "mi_debug_set_mg_planes": (_I, [_I]),
    "mi_debug_set_mg_lean": (_I, [_I]),
    "mi_debug_set_mg_nosync": (_I, [_I]),
    "mi_debug_set_mg_deg_cap": (_I, [_I]),
    "mi_gbatch_graph_status": (_I, [_P, _P, C.POINTER(_I), _P]),
    "mi_gemnet_tap": (_I, [_P, C.c_char_p, _P, _L, C.POINTER(_L), _P]),
    "mi_mg_sample_marginal": (_I, [_P, C.POINTER(MGCorruption), _P, _P, _P, _P, _U64, _U32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mi_mg_sampler_init": (_I, [_P, C.POINTER(MGCorruption), _U64, _P, _P, _P, _P, _P, _P]),
    "mi_mg_sampler_run": (_I, [_P, _P, C.POINTER(MGCorruption), _I, _I, _I, C.POINTER(C.c_float), _U64, C.POINTER(MGSamplerNoise), _P, _P, _P,
                               _P, _P, _P]),
    "mi_set_gemm_mode": (_I, [_I]),
    "mi_net_set_edge_mode": (_I, [_P, _I]),
    "mi_debug_gemm": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "mi_saturation_events": (_I, [C.POINTER(_L), _I]),
    "mi_profile_enable": (_I, [_P, _I]),
    "mi_profile_read": (_I, [_P, C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Load the shared library (once) and attach the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: both must share ONE HIP runtime instance (torch bundles its own libamdhip64);
    # device pointers and streams cross this boundary.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"matinvent_amd: HIP library not found at {LIB_PATH}; build it with `python -m matinvent_amd.build` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


MI_EINVAL, MI_EHIP, MI_ENOMEM, MI_ESTATE, MI_ECAPACITY = -1, -2, -3, -4, -5   # include/matinvent_hip.h


class MIError(RuntimeError):
    """A failed library call; `code` is the C ABI's return value (MI_EINVAL / MI_EHIP / MI_ENOMEM / MI_ESTATE / MI_ECAPACITY)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mi_last_error().decode(errors="replace")
        raise MIError(rc, f"matinvent_hip {what} failed (code {rc}): {msg}")


def saturation_events(reset: bool = True) -> int:
    """Number of fp32 -> fp16-plane conversions that had to clamp (or met NaN / inf) since the last reset, over every network and
    stream of the current device (mi_saturation_events; synchronises the device)."""
    n = C.c_int64()
    check(load().mi_saturation_events(C.byref(n), 1 if reset else 0), "mi_saturation_events")
    return int(n.value)


def check_saturation(where: str):
    """Raise if any operand left the range of the two-plane fp16 format since the last check: such results are finite but wrong,
    and must never be handed on silently."""
    n = saturation_events(reset=True)
    if n:
        raise FloatingPointError(
            f"{where}: {n} operand conversions saturated the two-plane fp16 format (values beyond 65504 / scale, or NaN / inf "
            "upstream): the results are outside the stated fp32-class tolerance.  Rebuild with MI_EXTRA_FLAGS=-DMI_PLANES_FP16=0 "
            "(three bf16 planes, no range limit) or use --path f32-gemm")
