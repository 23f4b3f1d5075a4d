"""``model.net`` plugin: HIP-backed drop-in for the reference ``WaveNetNoise`` (BASELINE configs[4]).

Contract kept (reference: src/models/backbones/wavenet.py:153-180):
  * constructor kwargs ``residual_channels=256, residual_layers=36, dilation_cycle=12``,
  * ``state_dict()`` keys / shapes identical to the reference, including the custom ``WeightNorm``'s 0-dim ``weight_g`` and
    ``weight_v`` registered after ``bias`` (:15-55), so reference checkpoints strict-load,
  * ``forward(audio[B, T], diffusion_step[B]) -> [B, 1, T]``.

The adapter THIS BUILD adds (the reference has none: its ``forward`` rejects the ``cond_drop_prob=`` / ``classes=`` keyword
arguments ``Diffusion.denoise_fn`` always passes, diffusion.py:50, so no reference caller sits above ``forward``):
``forward`` also takes the EDM wrapper's ``[B, 1, T]`` input and ignores conditioning keyword arguments, which lets
``EluDiffusion.denoise_fn`` and every sampler of this package drive the network like the U-Net -- on the device that is one
``adf_denoise`` / ``adf_sampler_run`` call on a handle made by ``adf_wavenet_create``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .config import WaveNetConfig
from .net import HipNet, _DTYPES
from .weights import wavenet_param_specs


def _init_like_reference(name: str, shape, kind: str) -> torch.Tensor:
    t = torch.empty(shape, dtype=torch.float32)
    if name.startswith("output_projection.conv."):
        return t.zero_()                                  # ZeroConv1d, wavenet.py:57-66
    if kind == "wn_g":
        return torch.tensor(1.0)                          # replaced below by ||w|| of the kaiming-initialised weight (:29)
    if kind == "conv_w":
        nn.init.kaiming_normal_(t)                        # :75
        return t
    if kind == "linear_w":
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
        return t
    return t.uniform_(-0.05, 0.05)


class WaveNetNoise(HipNet):
    """HIP-backed ``WaveNetNoise``.  Extra kwarg: ``compute_dtype`` in {"fp32", "bf16"} (bf16 = MFMA kernels, needs
    ``residual_channels=256``)."""

    def __init__(self, residual_channels: int = 256, residual_layers: int = 36, dilation_cycle: int = 12,
                 compute_dtype: str = "fp32"):
        super().__init__()
        if compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_DTYPES)}")
        if residual_channels % 32 or not 32 <= residual_channels <= 512:
            raise ValueError("residual_channels must be a multiple of 32 in [32, 512]")
        if _DTYPES[compute_dtype] == _DTYPES["bf16"] and residual_channels not in (64, 128, 256):
            raise ValueError("the bf16 (MFMA) kernels are built for residual_channels = 64, 128 or 256; use compute_dtype='fp32' otherwise")
        self.compute_dtype = compute_dtype
        self.cfg = WaveNetConfig(residual_channels=residual_channels, residual_layers=residual_layers, dilation_cycle=dilation_cycle)
        specs = wavenet_param_specs(self.cfg)
        for name, (shape, kind) in specs.items():
            self._register(name, nn.Parameter(_init_like_reference(name, shape, kind)))
        with torch.no_grad():                             # WeightNorm._reset (:24-42): g = ||w||, v = w / g
            params = dict(self.named_parameters())
            for name in specs:
                if name.endswith("weight_g"):
                    v = params[name[:-1] + "v"]
                    g = torch.norm(v)
                    params[name].copy_(g)
                    v.div_(g)

    @classmethod
    def from_config(cls, cfg: WaveNetConfig, compute_dtype: str = "fp32") -> "WaveNetNoise":
        return cls(compute_dtype=compute_dtype, **cfg.to_kwargs())

    def forward(self, audio: torch.Tensor, diffusion_step: torch.Tensor, **_ignored) -> torch.Tensor:
        if torch.is_grad_enabled() and audio.requires_grad:
            raise NotImplementedError("the HIP WaveNetNoise is an inference path (no backward); call it under torch.no_grad()")
        if audio.ndim == 3 and audio.shape[1] == 1:       # the EDM wrapper's [B, 1, T]
            x = audio
        elif audio.ndim == 2:                             # the reference's [B, T] (:171 unsqueezes it)
            x = audio.unsqueeze(1)
        else:
            raise ValueError("audio must be shaped [B, T] or [B, 1, T]")
        hd = self.native(x.device)
        xin = x.detach().to(torch.float32).contiguous()
        tin = diffusion_step.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if tin.numel() != xin.shape[0]:
            raise ValueError("diffusion_step must have one entry per batch element")
        with torch.cuda.device(x.device):
            return hd.net_forward(xin, tin).to(audio.dtype)
