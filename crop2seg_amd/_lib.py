"""ctypes binding of libc2s_hip.so (the C ABI declared in include/c2s_hip.h).

The product path has NO CPU fallback: if the library cannot be loaded, `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C2S_LIB", os.path.join(HERE, "libc2s_hip.so"))   # C2S_LIB: an alternative build (A/B runs)

c_float_p = C.c_void_p   # device pointers are passed as integers (tensor.data_ptr())
c_int_p = C.c_void_p

PAD_ZEROS, PAD_REFLECT = 0, 1
SRC_F32, SRC_I16, SRC_U16 = 0, 1, 2
NORM_GROUP, NORM_BATCH = 0, 1


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "N", "C0", "C1", "Hin", "Win", "Cout", "CoutP", "Hout", "Wout", "OutH", "OutW", "KH", "KW", "S",
        "pad_y", "pad_x", "pad_mode", "osy", "osx", "ooy", "oox", "accumulate", "reflect_adjoint")]


class WgradDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "N", "C0", "C1", "Hin", "Win", "Cout", "Hout", "Wout", "KH", "KW", "S", "pad_y", "pad_x", "pad_mode",
        "nslices")]


class NormDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("kind", C.c_int), ("groups", C.c_int),
                ("training", C.c_int), ("eps", C.c_float), ("momentum", C.c_float)]


class LtaeDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("n_head", C.c_int),
                ("d_model", C.c_int), ("eps", C.c_float), ("dropout_p", C.c_float), ("seed", C.c_uint64),
                ("keep", C.c_void_p), ("seed_dev", C.c_void_p), ("keep_bits", C.c_void_p)]


class AggDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "T", "C", "H", "W", "n_head", "h", "w")]


P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float
SZ = C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/c2s_hip.h
SIGNATURES = {
    "c2s_abi_version": (I, []),
    "c2s_last_error": (C.c_char_p, []),
    "c2s_init": (I, [I]),
    "c2s_device_cus": (I, []),
    "c2s_pack_weights": (I, [P, P, I, I, I, I, L, L, C.POINTER(I), P]),
    "c2s_pack_job_bytes": (SZ, []),
    "c2s_pack_job_fill": (I, [P, P, P, I, I, I, I, L, L, I, C.POINTER(I), I]),
    "c2s_pack_job_blocks": (I, [I, I, I, I]),
    "c2s_pack_batch": (I, [P, I, I, P]),
    "c2s_conv_igemm": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P, P]),
    "c2s_conv_xpair": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P]),
    "c2s_conv3x3_smallcin_supported": (I, [C.POINTER(ConvDesc)]),
    "c2s_conv3x3_smallcin": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P]),
    "c2s_winograd_packed_floats": (SZ, [I, I]),
    "c2s_pack_weights_winograd": (I, [P, P, I, I, I, L, L, C.POINTER(I), P]),
    "c2s_conv3x3_winograd": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P, P]),
    "c2s_winograd16_packed_floats": (SZ, [I, I]),
    "c2s_pack_weights_winograd16": (I, [P, P, I, I, I, L, L, C.POINTER(I), P]),
    "c2s_conv3x3_winograd16_supported": (I, [C.POINTER(ConvDesc)]),
    "c2s_conv3x3_winograd16": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P, P]),
    "c2s_s2wino_packed_floats": (SZ, [I, I]),
    "c2s_pack_weights_s2wino": (I, [P, P, I, I, I, L, L, C.POINTER(I), P]),
    "c2s_conv4x4s2_winograd_supported": (I, [C.POINTER(ConvDesc)]),
    "c2s_conv4x4s2_winograd": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P]),
    "c2s_s2dgrad_packed_floats": (SZ, [I, I]),
    "c2s_pack_weights_s2dgrad": (I, [P, P, I, I, I, L, L, C.POINTER(I), P]),
    "c2s_conv4x4s2_dgrad_winograd_supported": (I, [C.POINTER(ConvDesc)]),
    "c2s_conv4x4s2_dgrad_winograd": (I, [C.POINTER(ConvDesc), P, P, P, P, P]),
    "c2s_bf16x3_packed_elems": (SZ, [I, I]),
    "c2s_pack_weights_bf16x3": (I, [P, P, P, I, I, I, I, L, L, C.POINTER(I), P]),
    "c2s_conv3x3_bf16x3": (I, [C.POINTER(ConvDesc), P, P, P, P, P, P, P, P]),
    "c2s_bf16x3_set_single_product": (None, [I]),
    "c2s_wgrad_workspace_floats": (SZ, [C.POINTER(WgradDesc)]),
    "c2s_conv_wgrad": (I, [C.POINTER(WgradDesc), P, P, P, P, SZ, P, P]),
    "c2s_wgrad_algorithms": (I, [I, I]),
    "c2s_wgrad_reduce": (I, [C.POINTER(WgradDesc), P, P, L, L, C.POINTER(I), I, P]),
    "c2s_wgrad_reduce_job_bytes": (SZ, []),
    "c2s_wgrad_reduce_job_blocks": (I, [C.POINTER(WgradDesc)]),
    "c2s_wgrad_reduce_job_fill": (I, [P, C.POINTER(WgradDesc), P, P, L, L, C.POINTER(I), I, I]),
    "c2s_wgrad_reduce_batch": (I, [P, I, I, P]),
    "c2s_dwconv_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "c2s_dwconv_dgrad": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "c2s_dwconv_wgrad": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "c2s_norm_workspace_floats": (SZ, [C.POINTER(NormDesc)]),
    "c2s_norm_fwd": (I, [C.POINTER(NormDesc), P, P, P, P, P, P, P, P, P, P, I, P, SZ, P, F, P]),
    "c2s_norm_bwd": (I, [C.POINTER(NormDesc), P, P, P, P, P, I, P, P, P, P, P, SZ, P, P]),
    "c2s_norm_bwd_params": (I, [C.POINTER(NormDesc), P, P, P, P, P, P]),
    "c2s_norm_onepass_sync_bytes": (SZ, [C.POINTER(NormDesc), I]),
    "c2s_norm_fwd_onepass": (I, [C.POINTER(NormDesc), P, P, P, P, P, P, P, P, P, P, I, P, F, P, SZ, P]),
    "c2s_norm_bwd_onepass": (I, [C.POINTER(NormDesc), P, P, P, P, P, I, P, P, P, P, P, SZ, P, P, SZ, P]),
    "c2s_se_workspace_floats": (SZ, [I, I, I]),
    "c2s_channel_bias_add": (I, [P, P, P, I, I, I, P]),
    "c2s_se_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, F, P, SZ, P]),
    "c2s_se_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, P, I, I, I, P, SZ, P]),
    "c2s_frame_flags": (I, [P, P, I, L, F, P]),
    "c2s_ltae_attn_fwd": (I, [C.POINTER(LtaeDesc)] + [P] * 13 + [P]),
    "c2s_positional_table": (I, [P, P, L, C.c_float, P]),
    "c2s_ltae_fold_fwd": (I, [P] * 9 + [I, I, P]),
    "c2s_ltae_fold_bwd": (I, [P] * 16 + [I, I, I, P, SZ, P]),
    "c2s_ltae_fold_bwd_workspace_floats": (SZ, []),
    "c2s_ltae_pe_table": (I, [I, P, P, F, P, P, P, P, P, I, P]),
    "c2s_ltae_pe_abs_add": (I, [P, P, P, P, P, I, P]),
    "c2s_ltae_pe_abs_bwd": (I, [P, P, P, P, I, P]),
    "c2s_ltae_pe_fwd": (I, [P, P, P, P, P, I, I, I, I, P]),
    "c2s_ltae_pe_gattn": (I, [P, P, P, P, I, I, I, P]),
    "c2s_ltae_pe_bwd": (I, [I] + [P] * 15 + [I, I, I, P]),
    "c2s_ltae_fwd_workspace_floats": (SZ, [C.POINTER(LtaeDesc)]),
    "c2s_ltae_attn_fwd_ws": (I, [C.POINTER(LtaeDesc)] + [P] * 13 + [P, SZ, P]),
    "c2s_ltae_uses_streaming": (I, [C.POINTER(LtaeDesc)]),
    "c2s_ltae_fwd_path": (I, [C.POINTER(LtaeDesc)]),
    "c2s_ltae_attn_optional": (I, [C.POINTER(LtaeDesc)]),
    "c2s_ltae_bwd_workspace_floats": (SZ, [C.POINTER(LtaeDesc)]),
    "c2s_ltae_attn_bwd": (I, [C.POINTER(LtaeDesc)] + [P] * 22 + [SZ, P]),
    "c2s_dropout_nchw": (I, [P, P, I, I, I, F, C.c_uint64, P, P, P]),
    "c2s_pixel_gn_fwd": (I, [P, P, P, P, P, I, I, I, I, F, P]),
    "c2s_pixel_gn_bwd_workspace_floats": (SZ, [I, I, I]),
    "c2s_pixel_gn_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, P, SZ, P]),
    "c2s_attn_head_mean": (I, [P, P, I, L, P]),
    "c2s_attn_head_mean_bwd": (I, [P, P, I, L, I, P]),
    "c2s_frame_mean_weights": (I, [P, P, I, I, I, P]),
    "c2s_temporal_aggregate_fwd": (I, [C.POINTER(AggDesc), P, P, P, P, P]),
    "c2s_temporal_aggregate_bwd_workspace_floats": (SZ, [C.POINTER(AggDesc)]),
    "c2s_temporal_aggregate_bwd": (I, [C.POINTER(AggDesc), P, P, P, P, P, I, P, P, SZ, P]),
    "c2s_cross_entropy_workspace_floats": (SZ, [I, I]),
    "c2s_cross_entropy": (I, [P, P, P, P, P, I, I, I, F, C.c_longlong, P, SZ, P]),
    "c2s_metrics_update": (I, [P, P, P, P, P, P, I, I, I, P]),
    "c2s_confusion_add": (I, [P, P, P, L, I, P]),
    "c2s_loss_meter_add": (I, [P, P, P]),
    "c2s_boundary_target": (I, [P, P, I, I, I, P]),
    "c2s_region_relabel": (I, [P, P, I, I, I, I, C.c_longlong, P]),
    "c2s_focal_ce_workspace_floats": (SZ, []),
    "c2s_focal_ce": (I, [P, P, P, P, I, I, I, F, C.c_longlong, I, P, SZ, P]),
    "c2s_focal_ce_ex": (I, [P, P, P, P, P, I, I, I, F, C.c_longlong, I, I, P, SZ, P]),
    "c2s_smooth_ce_workspace_floats": (SZ, []),
    "c2s_smooth_ce": (I, [P, P, P, P, P, P, I, I, I, I, F, C.c_longlong, I, P, SZ, P]),
    "c2s_smooth_ce_ex": (I, [P, P, P, P, P, P, P, I, I, I, I, F, C.c_longlong, I, I, P, SZ, P]),
    "c2s_collate_series": (I, [P, I, P, P, P, P, P, I, I, I, I, I, C.POINTER(I), C.POINTER(F), C.POINTER(F), F, P]),
    "c2s_collate_series_ndvi": (I, [P, I, P, P, P, P, P, I, I, I, I, I, C.POINTER(I), C.POINTER(F), C.POINTER(F), F, I, I, P]),
    "c2s_softmax_stitch": (I, [P, P, P, I, I, I, I, I, I, I, I, P]),
    "c2s_adam_flat": (I, [P, P, P, P, L, F, F, F, F, I, P, F, P]),
    "c2s_fill": (I, [P, L, F, P]),
    "c2s_add_inplace": (I, [P, P, L, P]),
}

_LIB: Optional[C.CDLL] = None


class C2SError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libc2s_hip.so (once).  Raises if it is missing: there is no fallback path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise C2SError(
                f"{LIB_PATH} not found: build it with `python -m crop2seg_amd.build` "
                "(or __graft_entry__.build()); crop2seg_amd has no CPU/PyTorch fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


_INITED = set()


def init_device(index: int) -> None:
    """c2s_init() for HIP device `index` (must be the current device): raises the kernels' dynamic-LDS limits and caches
    the CU count before any launch -- in particular before a hipGraph capture."""
    if index in _INITED:
        return
    check(lib().c2s_init(index), "c2s_init")
    _INITED.add(index)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().c2s_last_error().decode(errors="replace")
        raise C2SError(f"{what} failed (code {rc}): {msg}")
