"""ctypes binding of libvdmi.so (declarations mirror include/vdmi.h).

There is no fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# Precision of the kernels' storage and matrix operands, one per process: "bf16" (default; libvdmi.so) or "f16"
# (VDMI_PRECISION=f16; libvdmi_f16.so: the same kernels built with IEEE half in place of bfloat16 - 11 significant
# bits instead of 8 at the same speed class, for loops whose final sample must stay within 2e-2 of fp32).
PRECISION = os.environ.get("VDMI_PRECISION", "bf16").lower()
if PRECISION not in ("bf16", "f16"):
    raise ValueError(f"VDMI_PRECISION={PRECISION!r}: expected 'bf16' or 'f16'")
ACT_DTYPE = torch.float16 if PRECISION == "f16" else torch.bfloat16
LIB_PATH = os.environ.get("VDMI_LIB_PATH") or os.path.join(_HERE, "libvdmi_f16.so" if PRECISION == "f16" else "libvdmi.so")

MAX_TAPS = 16
EPI_PLAIN, EPI_NORM_SILU, EPI_QKV_ATTN, EPI_QKV_SLA = 0, 1, 2, 3


class TapPass(C.Structure):
    _fields_ = [
        ("src0", C.c_void_p), ("src1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("stride", C.c_int32), ("ntaps", C.c_int32),
        ("dy", C.c_int8 * MAX_TAPS), ("dx", C.c_int8 * MAX_TAPS),
        ("w", C.c_void_p),
        ("w_group_stride", C.c_int64),
        ("w_group_frames", C.c_int32), ("phases", C.c_int32),
        ("w_phase_stride", C.c_int64),
        ("space_to_depth", C.c_int32), ("reserved", C.c_int32),
    ]


class TapGemmArgs(C.Structure):
    _fields_ = [
        ("main", TapPass), ("res", TapPass),
        ("has_res", C.c_int32),
        ("res_identity", C.c_void_p),
        ("ld_res", C.c_int32),
        ("res_bias", C.c_void_p),
        ("frames", C.c_int32), ("frames_per_batch", C.c_int32),
        ("grid_h", C.c_int32), ("grid_w", C.c_int32),
        ("out_mul", C.c_int32), ("out_off_y", C.c_int32), ("out_off_x", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32),
        ("out", C.c_void_p),
        ("ldo", C.c_int32), ("n", C.c_int32), ("epilogue", C.c_int32),
        ("bias", C.c_void_p), ("gain", C.c_void_p), ("shift", C.c_void_p),
        ("gain_bstride", C.c_int64), ("shift_bstride", C.c_int64),
        ("norm_scale", C.c_float),
        ("colsum", C.c_void_p),
        ("ln_eps", C.c_float), ("q_scale", C.c_float),
        ("rot_cos", C.c_void_p), ("rot_sin", C.c_void_p),
        ("rot_frames", C.c_int32), ("rot_hw", C.c_int32),
        ("out2", C.c_void_p),
        ("sla_hw", C.c_int32), ("tile_bn", C.c_int32),
        ("q_cols", C.c_int32), ("act", C.c_int32),
        ("k_split", C.c_int32), ("out_split_stride", C.c_int64),
    ]


class DdpmCoefs(C.Structure):
    _fields_ = [
        ("sqrt_recip_alphas_cumprod", C.c_void_p),
        ("sqrt_recipm1_alphas_cumprod", C.c_void_p),
        ("posterior_mean_coef1", C.c_void_p),
        ("posterior_mean_coef2", C.c_void_p),
        ("posterior_log_variance_clipped", C.c_void_p),
    ]


class PackJob(C.Structure):
    """include/vdmi.h: vdmi_pack_job"""
    _fields_ = [("w", C.c_void_p), ("in_scale", C.c_void_p), ("gather", C.c_void_p), ("out", C.c_void_p),
                ("n", C.c_int), ("npad", C.c_int), ("cin0", C.c_int), ("cin1", C.c_int), ("ntaps", C.c_int),
                ("bkc", C.c_int), ("transposed", C.c_int)]


class VdmiError(RuntimeError):
    pass


_lib = None

_SIGNATURES = {
    "vdmi_strerror": (C.c_char_p, [C.c_int]),
    "vdmi_version": (C.c_int, []),
    "vdmi_tapgemm_tile": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vdmi_pack_taps_elems": (C.c_size_t, [C.c_int] * 6),
    "vdmi_pack_taps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p]),
    "vdmi_tapgemm": (C.c_int, [C.POINTER(TapGemmArgs), C.c_void_p]),
    "vdmi_wsconv_enable": (None, [C.c_int]),
    "vdmi_wsconv_covers": (C.c_int, [C.c_void_p]),
    "vdmi_dconv_enable": (None, [C.c_int]),
    "vdmi_dconv_covers": (C.c_int, [C.c_void_p]),
    "vdmi_attn_core": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vdmi_pack_tattn_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "vdmi_pack_tattn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_tattn_fused": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "vdmi_tattn_fused_wide": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "vdmi_sattn_fused_wide": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_void_p]),
    "vdmi_sla_fused_wide": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_void_p]),
    "vdmi_pack_sla_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "vdmi_pack_sla": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_sla_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "vdmi_sla_fused": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_void_p]),
    "vdmi_sla_context": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p]),
    "vdmi_init_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "vdmi_final_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "vdmi_time_embed": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int] +
                        [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdmi_gather_scale_shift": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "vdmi_block_scale_shift": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 +
                               [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdmi_relpos_buckets": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_groupnorm_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int]),
    "vdmi_groupnorm_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "vdmi_groupnorm_silu_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_float, C.c_void_p]),
    "vdmi_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "vdmi_temporal_dwconv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdmi_groupnorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "vdmi_groupnorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "vdmi_temporal_dwconv_wgrad_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int]),
    "vdmi_temporal_dwconv_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                             C.c_int, C.c_int, C.c_void_p]),
    "vdmi_xattn_kv_elems": (C.c_size_t, [C.c_int] * 4),
    "vdmi_xattn_context": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_xattn_pack_kv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_xattn_pack_kv_train": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_xattn_core_train_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                            C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_xattn_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    "vdmi_xattn_core_train_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vdmi_xattn_core": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                  C.c_void_p]),
    "vdmi_rmsnorm_silu": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "vdmi_rmsnorm_silu_sum": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_void_p,
                                        C.c_int, C.c_void_p]),
    "vdmi_ddpm_step_rng": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DdpmCoefs), C.c_void_p, C.c_int,
                                     C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "vdmi_ddpm_step_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DdpmCoefs), C.c_void_p, C.c_int,
                                      C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vdmi_ddpm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DdpmCoefs),
                                 C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]),
    "vdmi_dynamic_threshold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DdpmCoefs), C.c_void_p, C.c_int,
                                         C.c_int64, C.c_float, C.c_void_p]),
    "vdmi_ddpm_step_thres": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DdpmCoefs), C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "vdmi_pack_taps_device": (C.c_int, [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]),
    "vdmi_pack_taps_batch_table_bytes": (C.c_size_t, [C.c_int]),
    "vdmi_pack_taps_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vdmi_rows_transpose_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int64]),
    "vdmi_rows_transpose": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "vdmi_conv_wgrad_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "vdmi_conv_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p] * 4),
    "vdmi_rows_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int64]),
    "vdmi_rows_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 4 +
                        [C.c_int64, C.c_void_p, C.c_void_p]),
    "vdmi_conv_wgrad_rows_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "vdmi_conv_wgrad_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 +
                             [C.c_void_p, C.c_void_p]),
    "vdmi_strided_wgrad_rows_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "vdmi_strided_wgrad_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 +
                                [C.c_void_p, C.c_void_p]),
    "vdmi_conv_wgrad_gather_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "vdmi_conv_wgrad_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]),
    "vdmi_rmsnorm_silu_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdmi_rmsnorm_silu_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int]),
    "vdmi_rmsnorm_silu_bwd_ws": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float] +
                                 [C.c_void_p] * 6),
    "vdmi_colsum": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "vdmi_colsum_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "vdmi_colsum_ws": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdmi_layernorm_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "vdmi_layernorm_rows_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "vdmi_tattn_core_fits": (C.c_int, [C.c_int] * 4),
    "vdmi_tattn_core_fwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "vdmi_tattn_core_bwd": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 4 + [C.c_float, C.c_void_p, C.c_void_p]),
    "vdmi_tattn_core_bwd_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "vdmi_tattn_core_mfma_enable": (None, [C.c_int]),
    "vdmi_sla_train_partial_bytes": (C.c_size_t, [C.c_int] * 3),
    "vdmi_sla_train_combined_bytes": (C.c_size_t, [C.c_int] * 2),
    "vdmi_sla_train_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] +
                              [C.c_int] * 4 + [C.c_void_p]),
    "vdmi_sla_train_rows_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "vdmi_sla_train_rows_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "vdmi_sla_rows_mfma_enable": (None, [C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """The loaded library; raises if libvdmi.so has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VdmiError(
                f"{LIB_PATH} is missing: build it with `make -C video_diffusion_amd/csrc` "
                "(or __graft_entry__.build()). There is no non-HIP fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(status: int, what: str = "vdmi call"):
    if status != 0:
        msg = lib().vdmi_strerror(status).decode()
        raise VdmiError(f"{what} failed: {msg} (status {status})")
