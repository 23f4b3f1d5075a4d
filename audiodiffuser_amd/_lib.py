"""ctypes binding of ``libadf_hip.so`` (C ABI declared in include/audiodiffuser_amd.h).

There is no fallback: if the library is missing or cannot be loaded the import of any
product entry point raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .config import UNet1dConfig

HERE = os.path.dirname(os.path.abspath(__file__))
# ADF_HIP_LIB: experiment switch only -- another build of the SAME HIP library (tools/build_variant.sh), so that two
# builds can be timed inside one gpurun call; there is no non-HIP implementation to point it at.
LIB_PATH = os.environ.get("ADF_HIP_LIB") or os.path.join(HERE, "libadf_hip.so")
ADF_MAX_LAYERS = 12
DTYPE_F32, DTYPE_BF16, DTYPE_F32X3 = 0, 1, 2
FLAG_SEPARATE_GN_STATS = 1
SAMPLER_EDM, SAMPLER_EDM_ALPHA, SAMPLER_DPM_MULTISTEP, SAMPLER_DPM2, SAMPLER_ADPM2 = 0, 1, 2, 3, 4
SAMPLER_LMS, SAMPLER_DPM_SINGLESTEP, SAMPLER_DPM2M, SAMPLER_UNIPC, SAMPLER_ADPMPP2S = 5, 6, 7, 8, 9


class AdfNetConfig(C.Structure):
    _fields_ = [
        ("channels", C.c_int32), ("num_filters", C.c_int32), ("window_length", C.c_int32), ("stride", C.c_int32),
        ("in_channels", C.c_int32), ("out_channels", C.c_int32),
        ("resnet_groups", C.c_int32), ("kernel_multiplier_downsample", C.c_int32),
        ("num_layers", C.c_int32),
        ("multipliers", C.c_int32 * (ADF_MAX_LAYERS + 1)),
        ("factors", C.c_int32 * ADF_MAX_LAYERS),
        ("num_blocks", C.c_int32 * ADF_MAX_LAYERS),
        ("attentions", C.c_int32 * ADF_MAX_LAYERS),
        ("attention_heads", C.c_int32), ("attention_multiplier", C.c_int32),
        ("use_skip_scale", C.c_int32), ("use_attention_bottleneck", C.c_int32),
        ("dtype", C.c_int32), ("flags", C.c_int32), ("num_classes", C.c_int32),
    ]


class AdfSamplerDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("num_steps", C.c_int32),
        ("s_tmin", C.c_float), ("s_tmax", C.c_float), ("s_churn", C.c_float), ("s_noise", C.c_float),
        ("use_heun", C.c_int32), ("alpha", C.c_float), ("order", C.c_int32), ("sigma_data", C.c_float),
        ("use_graph", C.c_int32), ("rho", C.c_float), ("eta", C.c_float), ("log_time_spacing", C.c_int32), ("eps_pred", C.c_int32),
        ("reflow", C.c_int32),
    ]


class AdfWaveNetConfig(C.Structure):
    """``adf_wavenet_config`` of include/audiodiffuser_amd.h."""
    _fields_ = [("residual_channels", C.c_int32), ("residual_layers", C.c_int32), ("dilation_cycle", C.c_int32),
                ("dim_in", C.c_int32), ("dim_mid", C.c_int32), ("dim_out", C.c_int32), ("dtype", C.c_int32)]


ADF_ADM_MAX_LEVELS = 8


class AdfAdmConfig(C.Structure):
    """``adf_adm_config`` of include/audiodiffuser_amd.h."""
    _fields_ = [("in_channels", C.c_int32), ("model_channels", C.c_int32), ("out_channels", C.c_int32), ("num_res_blocks", C.c_int32),
                ("n_mult", C.c_int32), ("channel_mult", C.c_int32 * ADF_ADM_MAX_LEVELS),
                ("n_attention_ds", C.c_int32), ("attention_ds", C.c_int32 * ADF_ADM_MAX_LEVELS),
                ("conv_resample", C.c_int32), ("num_heads", C.c_int32), ("num_head_channels", C.c_int32), ("use_scale_shift_norm", C.c_int32),
                ("resblock_updown", C.c_int32), ("use_new_attention_order", C.c_int32), ("num_classes", C.c_int32), ("dtype", C.c_int32)]


class AdfRunCounters(C.Structure):
    """``adf_run_counters`` of include/audiodiffuser_amd.h."""
    _fields_ = [("sampler_runs", C.c_int64), ("sampler_evals", C.c_int64), ("graph_captures", C.c_int64),
                ("graph_replays", C.c_int64), ("denoise_calls", C.c_int64), ("net_passes", C.c_int64)]


FLAG_NEAREST_UPSAMPLE = 2  # ADF_FLAG_NEAREST_UPSAMPLE
ABI_VERSION = 5          # ADF_ABI_VERSION of the header this binding was written against

EXPORTS = {
    # name: (restype, argtypes)
    "adf_abi_version": (C.c_int, []),
    "adf_get_counters": (C.c_int, [C.c_void_p, C.POINTER(AdfRunCounters)]),
    "adf_create": (C.c_int, [C.POINTER(AdfNetConfig), C.POINTER(C.c_void_p)]),
    "adf_wavenet_create": (C.c_int, [C.POINTER(AdfWaveNetConfig), C.POINTER(C.c_void_p)]),
    "adf_adm_create": (C.c_int, [C.POINTER(AdfAdmConfig), C.POINTER(C.c_void_p)]),
    "adf_set_image_shape": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "adf_set_dynamic_threshold": (C.c_int, [C.c_void_p, C.c_float]),
    "adf_destroy": (None, [C.c_void_p]),
    "adf_last_error": (C.c_char_p, [C.c_void_p]),
    "adf_num_weights": (C.c_int, [C.c_void_p]),
    "adf_weight_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "adf_weight_numel": (C.c_int64, [C.c_void_p, C.c_int]),
    "adf_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "adf_weights_missing": (C.c_int, [C.c_void_p]),
    "adf_set_condition": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "adf_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "adf_denoise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "adf_sampler_run": (C.c_int, [C.c_void_p, C.POINTER(AdfSamplerDesc), C.POINTER(C.c_float), C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "adf_sampler_nfe": (C.c_int, [C.POINTER(AdfSamplerDesc), C.POINTER(C.c_float), C.c_int]),
    "adf_debug_tap_shape": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "adf_debug_tap_copy": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p]),
    "adf_debug_dyn_threshold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    "adf_debug_tap_count": (C.c_int, [C.c_void_p]),
    "adf_debug_tap_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "adf_device_bytes": (C.c_int64, [C.c_void_p]),
    "adf_bench_resblock": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.c_void_p]),
    "adf_bench_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]),
    "adf_bench_wavenet_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """Load libadf_hip.so and bind every symbol of the public header; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m audiodiffuser_amd.build` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.adf_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, this binding was written against {ABI_VERSION}: rebuild the "
                           "library (python -m audiodiffuser_amd.build --force)")
    _lib = lib
    return lib


def make_config(cfg: UNet1dConfig, dtype: int, flags: int = 0) -> AdfNetConfig:
    cfg.validate()
    n = cfg.num_layers
    if n > ADF_MAX_LAYERS:
        raise ValueError("too many layers")
    c = AdfNetConfig()
    c.channels, c.num_filters, c.window_length, c.stride = cfg.channels, cfg.num_filters, cfg.window_length, cfg.stride
    c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
    c.resnet_groups, c.kernel_multiplier_downsample = cfg.resnet_groups, cfg.kernel_multiplier_downsample
    c.num_layers = n
    for i, m in enumerate(cfg.multipliers):
        c.multipliers[i] = int(m)
    for i in range(n):
        c.factors[i] = int(cfg.factors[i])
        c.num_blocks[i] = int(cfg.num_blocks[i])
        c.attentions[i] = 1 if cfg.attentions[i] else 0
    c.attention_heads, c.attention_multiplier = cfg.attention_heads, cfg.attention_multiplier
    c.use_skip_scale = 1 if cfg.use_skip_scale else 0
    c.use_attention_bottleneck = 1 if cfg.use_attention_bottleneck else 0
    c.dtype, c.flags = dtype, flags | (FLAG_NEAREST_UPSAMPLE if cfg.use_nearest_upsample else 0)
    c.num_classes = int(cfg.num_classes) if cfg.class_cond else 0
    return c


def make_adm_config(cfg, dtype: int) -> AdfAdmConfig:
    c = AdfAdmConfig()
    c.in_channels, c.model_channels, c.out_channels, c.num_res_blocks = cfg.in_channels, cfg.model_channels, cfg.out_channels, cfg.num_res_blocks
    if len(cfg.channel_mult) > ADF_ADM_MAX_LEVELS or len(cfg.attention_ds) > ADF_ADM_MAX_LEVELS:
        raise ValueError("too many levels")
    c.n_mult = len(cfg.channel_mult)
    for i, m in enumerate(cfg.channel_mult):
        if int(m) != m:
            raise ValueError("channel_mult entries must be integers")
        c.channel_mult[i] = int(m)
    c.n_attention_ds = len(cfg.attention_ds)
    for i, d in enumerate(cfg.attention_ds):
        c.attention_ds[i] = int(d)
    c.conv_resample, c.num_heads, c.num_head_channels = int(cfg.conv_resample), cfg.num_heads, cfg.num_head_channels
    c.use_scale_shift_norm, c.resblock_updown = int(cfg.use_scale_shift_norm), int(cfg.resblock_updown)
    c.use_new_attention_order = int(cfg.use_new_attention_order)
    c.num_classes = int(cfg.num_classes) if cfg.num_classes is not None else 0
    c.dtype = dtype
    return c


def make_wavenet_config(cfg, dtype: int) -> AdfWaveNetConfig:
    c = AdfWaveNetConfig()
    c.residual_channels, c.residual_layers, c.dilation_cycle = cfg.residual_channels, cfg.residual_layers, cfg.dilation_cycle
    c.dim_in, c.dim_mid, c.dim_out = cfg.dim_in, cfg.dim_mid, cfg.dim_out
    c.dtype = dtype
    return c


class AdfError(RuntimeError):
    pass


def check(lib: C.CDLL, handle, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.adf_last_error(handle)
        raise AdfError(f"{what}: {msg.decode() if msg else 'unknown error'}")
