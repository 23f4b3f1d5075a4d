"""``model.net`` plugin: HIP-backed drop-in for the reference ``UNet1dBase``.

Contract kept (reference: src/models/backbones/unet1d.py:818-893):
  * constructor kwargs of ``UNet1dBase`` / ``UNet1d`` (hydra ``_target_`` instantiation),
  * ``forward(x, t, classes=None, ..., cond_drop_prob=None, **kwargs) -> Tensor`` same shape as ``x``,
  * ``state_dict()`` keys/shapes identical to the reference, so Lightning strict-loads
    reference checkpoints (src/eval.py:73),
  * ``.parameters()`` non-empty (dtype probe, src/models/diffunet_complex_module.py:108).

The parameters are ordinary ``nn.Parameter`` s held in a module tree that reproduces the
reference key names; the compute happens in ``libadf_hip.so`` on packed copies of them that
are refreshed whenever a parameter changes.  PyTorch is only the memory container.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from .adm_config import ADMConfig
from .config import UNet1dConfig, WaveNetConfig
from .weights import param_specs

_DTYPES = {"fp32": _lib.DTYPE_F32, "float32": _lib.DTYPE_F32, "bf16": _lib.DTYPE_BF16, "bfloat16": _lib.DTYPE_BF16,
           "f32x3": _lib.DTYPE_F32X3}      # split-bf16 (UNet1dBase only): fp32 storage, bf16 hi + lo operands, 3 MFMAs per product


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class NativeHandle:
    """Owns one ``adf_handle`` (one device, one compute dtype)."""

    def __init__(self, cfg, dtype: str, flags: int = 0):
        self.lib = _lib.load_library()
        self.cfg = cfg
        self.dtype = dtype
        h = C.c_void_p()
        if isinstance(cfg, WaveNetConfig):
            c = _lib.make_wavenet_config(cfg, _DTYPES[dtype])
            rc, what = self.lib.adf_wavenet_create(C.byref(c), C.byref(h)), "adf_wavenet_create"
        elif isinstance(cfg, ADMConfig):
            c = _lib.make_adm_config(cfg, _DTYPES[dtype])
            rc, what = self.lib.adf_adm_create(C.byref(c), C.byref(h)), "adf_adm_create"
        else:
            c = _lib.make_config(cfg, _DTYPES[dtype], flags)
            rc, what = self.lib.adf_create(C.byref(c), C.byref(h)), "adf_create"
        if rc != 0:
            raise _lib.AdfError(what + ": " + self.lib.adf_last_error(None).decode())
        self.h = h
        self._loaded: Dict[str, tuple] = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.adf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str) -> None:
        _lib.check(self.lib, self.h, rc, what)

    def invalidate(self) -> None:
        """Forget what was uploaded: the next call re-packs every tensor."""
        self._loaded.clear()

    def sync_weights(self, named: Dict[str, torch.Tensor], device: torch.device) -> None:
        """Upload every tensor whose storage/version changed since the last call.  (A write through ``p.data`` or a raw
        pointer does not bump ``_version``: call ``UNet1dBase.invalidate_native()`` after such an update.)"""
        stream = _stream_ptr(device)
        for name, t in named.items():
            sig = (t.data_ptr(), t._version, t.device)
            if self._loaded.get(name) == sig:
                continue
            src = t.detach()
            if src.device != device or src.dtype != torch.float32 or not src.is_contiguous():
                src = src.to(device=device, dtype=torch.float32).contiguous()
            self.check(self.lib.adf_load_weight(self.h, name.encode(), C.c_void_p(src.data_ptr()), src.numel(), C.c_void_p(stream)),
                       f"adf_load_weight({name})")
            self._loaded[name] = sig
        missing = self.lib.adf_weights_missing(self.h)
        if missing:
            raise _lib.AdfError(f"{missing} state_dict tensors were not provided to the HIP library")

    # ---- class conditioning / classifier-free guidance state for the calls that follow -------
    def set_condition(self, classes: Optional[torch.Tensor], device: torch.device, null_labels: bool = False,
                      cond_scale: float = 1.0) -> None:
        """``classes``: int64 labels [B] (or None to clear).  Mirrors ``UNet1dBase.forward(classes=, cond_drop_prob=)``
        (cond_drop_prob 1 = ``null_labels``) and the ``cond_scale`` of ``Diffusion.denoise_fn`` (diffusion.py:49-54)."""
        if classes is None:
            self.check(self.lib.adf_set_condition(self.h, C.c_void_p(0), 0, 0, 1.0, C.c_void_p(_stream_ptr(device))), "adf_set_condition")
            return
        if not self.cfg.class_cond:
            raise ValueError("classes were given to a network built without class_cond=True")
        cl = classes.detach().to(device=device, dtype=torch.int64).reshape(-1).contiguous()
        if cl.numel() and (int(cl.min()) < 0 or int(cl.max()) >= self.cfg.num_classes):
            raise IndexError("class label out of range")           # nn.Embedding raises the same way
        self.check(self.lib.adf_set_condition(self.h, C.c_void_p(cl.data_ptr()), cl.numel(), 1 if null_labels else 0, float(cond_scale),
                                              C.c_void_p(_stream_ptr(device))), "adf_set_condition")

    # ---- compute entry points (all tensors fp32, contiguous, on the handle's device) ----
    def _length(self, x: torch.Tensor) -> int:
        """The C ABI's length argument: L for [B, C, L]; H * W for the 2-D U-Net's [B, C, H, W] (after telling the handle the shape)."""
        if x.ndim == 4:
            if not isinstance(self.cfg, ADMConfig):
                raise ValueError("a 4-D input needs the 2-D UNetModel")
            self.check(self.lib.adf_set_image_shape(self.h, int(x.shape[2]), int(x.shape[3])), "adf_set_image_shape")
            return int(x.shape[2] * x.shape[3])
        if isinstance(self.cfg, ADMConfig):
            raise ValueError("the 2-D UNetModel takes [B, C, H, W] inputs")
        return int(x.shape[-1])

    def net_forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        out = torch.empty((x.shape[0], self.cfg.out_channels) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
        self.check(self.lib.adf_net_forward(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()),
                                            x.shape[0], self._length(x), C.c_void_p(_stream_ptr(x.device))), "adf_net_forward")
        return out

    def set_dynamic_threshold(self, quantile: float) -> None:
        """Clipping of every later denoiser evaluation: 0 = clamp(-1, 1), q = EluDiffusion(dynamic_threshold=q) (components/utils.py:19-33)."""
        self.check(self.lib.adf_set_dynamic_threshold(self.h, float(quantile)), "adf_set_dynamic_threshold")

    def denoise(self, x: torch.Tensor, sigma_data: float, sigma: Optional[float] = None,
                sigmas: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = torch.empty_like(x)
        sp = C.c_void_p(sigmas.data_ptr()) if sigmas is not None else C.c_void_p(0)
        self.check(self.lib.adf_denoise(self.h, C.c_void_p(x.data_ptr()), sp, float(sigma if sigma is not None else 0.0),
                                        float(sigma_data), C.c_void_p(out.data_ptr()), x.shape[0], self._length(x),
                                        C.c_void_p(_stream_ptr(x.device))), "adf_denoise")
        return out

    def sampler_run(self, desc: "_lib.AdfSamplerDesc", sigmas_host: torch.Tensor, noise: torch.Tensor,
                    injected: Optional[torch.Tensor]) -> torch.Tensor:
        out = torch.empty_like(noise)
        sg = sigmas_host.detach().to("cpu", torch.float32).contiguous()
        arr = (C.c_float * sg.numel())(*sg.tolist())
        n_inj = 0
        if injected is not None:
            if (injected.ndim != noise.ndim + 1 or tuple(injected.shape[1:]) != tuple(noise.shape) or injected.dtype != torch.float32
                    or injected.device != noise.device or not injected.is_contiguous()):
                raise ValueError("injected noise must be a contiguous fp32 [n, *noise.shape] tensor on the device of `noise`")
            n_inj = int(injected.shape[0])
        ip = C.c_void_p(injected.data_ptr()) if injected is not None else C.c_void_p(0)
        with torch.cuda.device(noise.device):
            self.check(self.lib.adf_sampler_run(self.h, C.byref(desc), arr, sg.numel(), C.c_void_p(noise.data_ptr()), ip, n_inj,
                                                C.c_void_p(out.data_ptr()), noise.shape[0], self._length(noise),
                                                C.c_void_p(_stream_ptr(noise.device))), "adf_sampler_run")
        return out

    def counters(self) -> Dict[str, int]:
        """What the device loop has done so far (``adf_get_counters``): sampler runs, evaluations, graph captures / replays."""
        c = _lib.AdfRunCounters()
        self.check(self.lib.adf_get_counters(self.h, C.byref(c)), "adf_get_counters")
        return {name: int(getattr(c, name)) for name, _ in c._fields_}

    def tap_names(self):
        n = self.lib.adf_debug_tap_count(self.h)
        return [self.lib.adf_debug_tap_name(self.h, i).decode() for i in range(n)]

    def tap(self, name: str, batch: int, device: torch.device) -> torch.Tensor:
        c, l = C.c_int(), C.c_int()
        self.check(self.lib.adf_debug_tap_shape(self.h, name.encode(), C.byref(c), C.byref(l)), "adf_debug_tap_shape")
        out = torch.empty((batch, c.value, l.value), device=device, dtype=torch.float32)      # 2-D nets: l = H * W of the tap's level
        self.check(self.lib.adf_debug_tap_copy(self.h, name.encode(), C.c_void_p(out.data_ptr()), C.c_void_p(_stream_ptr(device))),
                   "adf_debug_tap_copy")
        return out


def _init_like_reference(name: str, shape, kind: str) -> torch.Tensor:
    """Default initialisation equivalent to the reference's torch defaults."""
    t = torch.empty(shape, dtype=torch.float32)
    if kind in ("conv_w", "convT_w", "linear_w"):
        if name == "unet.to_out.to_out.weight":
            return t.zero_()                      # unet1d.py:619
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
        return t
    if kind == "bias":
        # torch default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); fan_in is not known here, keep it small
        return t.uniform_(-0.05, 0.05)
    if kind == "norm_w":
        return t.fill_(1.0)
    if kind == "norm_b":
        return t.zero_()
    if kind in ("fourier", "embed"):
        return t.normal_()
    raise ValueError(kind)


class HipNet(nn.Module):
    """What the HIP-backed ``model.net`` plugins share: a parameter tree under the reference's state_dict key names, one
    ``NativeHandle`` per (device, compute dtype) whose packed weights follow the parameters, and the ``cfg`` the denoise /
    sampler fast paths read (``class_cond``, ``out_channels``).  Subclasses set ``cfg``, ``compute_dtype``, ``native_flags``."""

    compute_dtype = "fp32"
    native_flags = 0

    # -- parameter tree with reference key names ------------------------------------
    def _register(self, dotted: str, p: nn.Parameter) -> None:
        mod = self
        parts = dotted.split(".")
        for part in parts[:-1]:
            if part not in mod._modules:
                mod.add_module(part, nn.Module())
            mod = mod._modules[part]
        mod.register_parameter(parts[-1], p)

    # -- native handle ---------------------------------------------------------------
    def native(self, device: torch.device) -> NativeHandle:
        if device.type != "cuda":
            raise RuntimeError(f"the HIP {type(self).__name__} only runs on a ROCm device ('cuda'); there is no CPU fallback")
        handles = self.__dict__.setdefault("_handles", {})
        key = (device.index if device.index is not None else torch.cuda.current_device(), self.compute_dtype)
        hd = handles.get(key)
        if hd is None:
            with torch.cuda.device(device):
                hd = NativeHandle(self.cfg, self.compute_dtype, self.native_flags)
            handles[key] = hd
        hd.sync_weights(dict(self.named_parameters()), device)
        return hd

    def invalidate_native(self) -> None:
        """Force a re-upload of all weights on the next call.  The staleness check keys on ``(data_ptr, _version, device)``;
        ``load_state_dict`` and in-place tensor ops bump ``_version``, but writes through ``p.data`` (some EMA / weight-swap
        code) or raw pointers do not -- call this after them."""
        for hd in self.__dict__.get("_handles", {}).values():
            hd.invalidate()

    def _load_from_state_dict(self, *args, **kwargs):       # (only parameters registered on this module itself, if any)
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_native()

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_native()
        return out


class UNet1dBase(HipNet):
    """HIP-backed ``UNet1dBase``.  Extra kwarg: ``compute_dtype`` in {"fp32", "bf16"}."""

    def __init__(self, channels: int, cond_drop_prob: float = 0.0, num_classes: Optional[int] = None,
                 class_embed_dim: Optional[int] = None, class_cond: bool = False, text_cond: bool = False,
                 max_text_len: Optional[int] = None, text_embed_dim: int = 768, text_cond_multiplier: Optional[int] = None,
                 use_self_text_cond: bool = False, use_condition_block: bool = False,
                 compute_dtype: str = "fp32", native_flags: int = 0, **kwargs):
        super().__init__()
        if text_cond or use_condition_block or use_self_text_cond:
            raise NotImplementedError("text / channel conditioning is outside the hot path (SURVEY.md 8f)")
        if class_cond and class_embed_dim is not None:
            raise NotImplementedError("class_embed_dim (embedding inputs instead of labels) is outside the hot path")
        if compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_DTYPES)}")
        out_channels = kwargs.pop("out_channels", None)
        self.cond_drop_prob = cond_drop_prob
        self.compute_dtype = compute_dtype
        self.native_flags = native_flags
        self.cfg = UNet1dConfig(channels=channels, cond_drop_prob=cond_drop_prob, class_cond=bool(class_cond),
                                num_classes=num_classes if class_cond else None, **kwargs)
        self.cfg.out_channels = out_channels if out_channels is not None else self.cfg.in_channels  # unet1d.py:607
        self.cfg.validate()
        self._specs = param_specs(self.cfg)
        for name, (shape, kind) in self._specs.items():
            self._register(name, nn.Parameter(_init_like_reference(name, shape, kind)))
        self._handles: Dict[tuple, NativeHandle] = {}

    @classmethod
    def from_config(cls, cfg: UNet1dConfig, compute_dtype: str = "fp32", native_flags: int = 0) -> "UNet1dBase":
        kw = cfg.to_kwargs()
        return cls(compute_dtype=compute_dtype, native_flags=native_flags, **kw)

    def forward(self, x: torch.Tensor, t: torch.Tensor, classes=None, text_embeds=None, text_mask=None,
                inj_embeddings=None, inj_channels=None, cond_drop_prob=None, **kwargs) -> torch.Tensor:
        if text_embeds is not None or inj_embeddings is not None or inj_channels is not None:
            raise NotImplementedError("text / injected conditioning inputs are outside the hot path (SURVEY.md 8f)")
        if torch.is_grad_enabled() and x.requires_grad:
            # the HIP path is inference only: its output carries no autograd history, so a loss built on it would fail at
            # backward() with an unrelated-looking error -- say so here instead
            raise NotImplementedError("the HIP UNet1dBase is an inference path (no backward); call it under torch.no_grad() "
                                      "or with inputs that do not require grad")
        hd = self.native(x.device)
        if self.cfg.class_cond:
            # unet1d.py:874-877: labels -> LabelEmbedder(classes, cond_drop_prob); the label mask is deterministic only
            # at cond_drop_prob 0 (keep all) and 1 (null embedding for all), the two values inference uses
            if classes is None:
                raise ValueError("a class-conditional UNet1dBase needs `classes`")
            cdp = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
            if cdp not in (0, 0.0, 1, 1.0):
                raise NotImplementedError("cond_drop_prob other than 0 or 1 draws a random label mask (training only)")
            hd.set_condition(classes, x.device, null_labels=bool(cdp), cond_scale=1.0)
        elif classes is not None:
            raise ValueError("classes were given to a network built without class_cond=True")
        xin = x.detach().to(torch.float32).contiguous()
        tin = t.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if tin.numel() != xin.shape[0]:
            raise ValueError("t must have one entry per batch element")
        with torch.cuda.device(x.device):
            return hd.net_forward(xin, tin).to(x.dtype)
