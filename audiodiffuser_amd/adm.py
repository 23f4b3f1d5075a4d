"""``model.net`` plugin: HIP-backed drop-in for the reference's ADM-style 2-D U-Net ``UNetModel`` (BASELINE configs[3], SURVEY.md 8f
row 3).

Contract kept (reference: src/models/backbones/unet2d_oai.py:382-635): the constructor kwargs, ``state_dict()`` keys / shapes
(reference checkpoints strict-load), ``forward(x[B, C, H, W], time[B], classes=None, cond_drop_prob=None) -> [B, out_channels, H, W]``.
On the device: ``use_scale_shift_norm=True``, conv resampling, no resblock up/down, either attention order, unconditional (BASELINE
config 4) or class-conditional with classifier-free guidance (``num_classes``; the shipped ``diffunet_complex_oai_sc09_cfg.yaml`` setting).
Other constructor variants raise (their oracle and fixtures exist: oracle/unet2d_oai.py).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .adm_config import ADMConfig, param_specs, structure
from .net import HipNet, _DTYPES

_ZERO_INIT = (".out_layers.3.", ".proj_out.", "out.2.")        # zero_module, unet2d_oai.py:227, :309, :599


def _init_like_reference(name: str, shape, kind: str) -> torch.Tensor:
    t = torch.empty(shape, dtype=torch.float32)
    if any(z in name or name.startswith(z) for z in _ZERO_INIT):
        return t.zero_()
    if kind in ("conv2d_w", "conv_w", "linear_w"):
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
        return t
    if kind == "norm_w":
        return t.fill_(1.0)
    if kind == "norm_b":
        return t.zero_()
    return t.uniform_(-0.05, 0.05)


class UNetModel(HipNet):
    """HIP-backed ``UNetModel``.  Extra kwarg: ``compute_dtype`` in {"fp32", "bf16"} (bf16 needs ``model_channels`` % 64 == 0)."""

    def __init__(self, image_size=256, in_channels=2, model_channels=128, out_channels=2, num_res_blocks=2, attention_resolutions="16",
                 dropout=0, channel_mult=(1, 2, 2, 4), conv_resample=True, num_classes=None, cond_drop_prob=0.0, use_checkpoint=False,
                 num_heads=8, num_head_channels=-1, use_scale_shift_norm=True, resblock_updown=False, use_new_attention_order=False,
                 class_embed_dim=None, compute_dtype: str = "fp32"):
        super().__init__()
        if compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_DTYPES)}")
        if class_embed_dim is not None:
            raise NotImplementedError("class_embed_dim (embedding inputs instead of labels) is outside the hot path")
        self.compute_dtype = compute_dtype
        self.cond_drop_prob = cond_drop_prob
        self.cfg = ADMConfig(image_size=image_size, in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                             num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, channel_mult=tuple(channel_mult),
                             conv_resample=conv_resample, num_classes=num_classes, num_heads=num_heads, num_head_channels=num_head_channels,
                             use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                             use_new_attention_order=use_new_attention_order)
        if compute_dtype in ("bf16", "bfloat16"):
            # the bf16 (MFMA) conv routes read K in 64-channel chunks from either source of a skip concat: say which layer breaks that HERE,
            # not as a generic "input channels must be a multiple of the 128-byte K chunk" at the first forward (WaveNetNoise does the same)
            st = structure(self.cfg)
            for layers in st.input_blocks[1:] + [st.middle] + st.output_blocks:
                for l in layers:
                    if l.cin % 64 or l.cout % 64:
                        raise ValueError(f"compute_dtype='bf16' needs every conv width to be a multiple of 64 channels: {l.pre} is "
                                         f"{l.cin} -> {l.cout} (model_channels={model_channels}, channel_mult={tuple(channel_mult)}); use compute_dtype='fp32'")
        for name, (shape, kind) in param_specs(self.cfg).items():
            self._register(name, nn.Parameter(_init_like_reference(name, shape, kind)))

    @classmethod
    def from_config(cls, cfg: ADMConfig, compute_dtype: str = "fp32") -> "UNetModel":
        return cls(compute_dtype=compute_dtype, **cfg.to_kwargs())

    def forward(self, x: torch.Tensor, time: torch.Tensor, classes: Optional[torch.Tensor] = None, cond_drop_prob=None, **_ignored) -> torch.Tensor:
        assert (classes is not None) == (self.cfg.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"                           # unet2d_oai.py:614-616
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("the HIP UNetModel is an inference path (no backward); call it under torch.no_grad()")
        if x.ndim != 4:
            raise ValueError("x must be shaped [B, C, H, W]")
        hd = self.native(x.device)
        if classes is not None:
            # LabelEmbedder(classes, cond_drop_prob): the label mask is deterministic only at cond_drop_prob 0 (keep all) and 1 (null embedding
            # for all), the two values inference uses (diffusion.py:50, :53)
            cdp = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
            if cdp not in (0, 0.0, 1, 1.0):
                raise NotImplementedError("cond_drop_prob other than 0 or 1 draws a random label mask (training only)")
            hd.set_condition(classes, x.device, null_labels=bool(cdp), cond_scale=1.0)
        xin = x.detach().to(torch.float32).contiguous()
        tin = time.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if tin.numel() != xin.shape[0]:
            raise ValueError("time must have one entry per batch element")
        with torch.cuda.device(x.device):
            return hd.net_forward(xin, tin).to(x.dtype)
