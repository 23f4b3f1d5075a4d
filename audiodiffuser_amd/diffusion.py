"""``model.diffusion`` plugin: EDM preconditioning + denoise wrapper
(reference: src/models/components/diffusion.py:15-63 ``Diffusion``, :217-257 ``EluDiffusion``).

``denoise_fn`` keeps the reference signature.  When ``net`` is the HIP ``UNet1dBase`` and the call
is the inference case with clamp clipping the whole thing -- c_in scaling, sigma embedding, U-Net,
(for a class-conditional net: label embedding and classifier-free guidance), c_skip/c_out combine,
clamp -- is one ``adf_denoise`` call.
For any other ``net`` (e.g. an unpickled reference module, diffunet_complex_module.py:239-242) the
same arithmetic is expressed with tensor ops around ``net(...)``; that branch exists for interface
compatibility and is not the accelerated path.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .net import HipNet, UNet1dBase


def _extend(x: Tensor, ndim: int) -> Tensor:
    return x.view(*x.shape, *((1,) * (ndim - x.ndim)))


class EluDiffusion(nn.Module):
    """Elucidated diffusion (EDM) preconditioning, table 1 of arXiv:2206.00364."""

    def __init__(self, sigma_data: float, dynamic_threshold: float = 0.0):
        super().__init__()
        self.sigma_data = sigma_data
        self.dynamic_threshold = dynamic_threshold

    # diffusion.py:232-241
    def get_scale_weights(self, sigmas: Tensor, ex_dim: int) -> Tuple[Tensor, ...]:
        sd = self.sigma_data
        c_noise = torch.log(sigmas) * 0.25
        s = _extend(sigmas, ex_dim)
        c_skip = (sd ** 2) / (s ** 2 + sd ** 2)
        c_out = s * sd * (sd ** 2 + s ** 2) ** -0.5
        c_in = (s ** 2 + sd ** 2) ** -0.5
        return c_skip, c_out, c_in, c_noise

    # diffusion.py:243-245
    def loss_weight(self, sigmas: Tensor) -> Tensor:
        return (sigmas ** 2 + self.sigma_data ** 2) * (sigmas * self.sigma_data) ** -2

    def _native_ok(self, net, inference: bool, cond_scale: float, kwargs: dict) -> bool:
        """The HIP fast path covers inference (clamp clipping or the dynamic threshold); the only conditioning kwarg it understands is
        ``classes`` (labels) on a class-conditional net, where ``cond_scale != 1`` is classifier-free guidance."""
        if not (isinstance(net, HipNet) and inference and 0.0 <= self.dynamic_threshold <= 1.0):
            return False
        extra = {k: v for k, v in kwargs.items() if v is not None}
        if net.cfg.class_cond:
            return set(extra) == {"classes"}
        return not extra and cond_scale == 1.0

    # diffusion.py:32-63
    def denoise_fn(self, x_noisy: Tensor, net: nn.Module = None, inference: bool = False, cond_scale: float = 1.0,
                   sigmas: Optional[Tensor] = None, sigma: Optional[float] = None, **kwargs) -> Tensor:
        assert (sigma is not None) ^ (sigmas is not None), "Either x or xs must be provided"   # components/utils.py:47
        if self._native_ok(net, inference, cond_scale, kwargs) and x_noisy.is_cuda:
            hd = net.native(x_noisy.device)
            hd.set_dynamic_threshold(self.dynamic_threshold)
            x = x_noisy.detach().to(torch.float32).contiguous()
            if net.cfg.class_cond:      # labels + guidance scale for this call (diffusion.py:49-54)
                hd.set_condition(kwargs["classes"], x.device, null_labels=False, cond_scale=float(cond_scale))
            with torch.cuda.device(x.device):
                if sigmas is not None:
                    sv = sigmas.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
                    return hd.denoise(x, self.sigma_data, sigmas=sv).to(x_noisy.dtype)
                return hd.denoise(x, self.sigma_data, sigma=float(sigma)).to(x_noisy.dtype)
        # ---- interface-compatibility branch: arbitrary `net` callable -------------------------
        if isinstance(net, HipNet) and inference and os.environ.get("ADF_REQUIRE_NATIVE", "0") not in ("", "0"):
            raise RuntimeError("denoise_fn: ADF_REQUIRE_NATIVE is set and this inference call on a HIP net would not be one adf_denoise call")
        b, device = x_noisy.shape[0], x_noisy.device
        if sigmas is None:
            sigmas = torch.full((b,), float(sigma), dtype=torch.float32, device=device)
        c_skip, c_out, c_in, c_noise = self.get_scale_weights(sigmas, x_noisy.ndim)
        if inference:
            pred = net(c_in * x_noisy, c_noise, cond_drop_prob=0.0, **kwargs)
            if cond_scale != 1.0:
                null = net(c_in * x_noisy, c_noise, cond_drop_prob=1.0, **kwargs)
                pred = null + (pred - null) * cond_scale
        else:
            pred = net(c_in * x_noisy, c_noise, **kwargs)
        den = c_skip * x_noisy + c_out * pred
        if self.dynamic_threshold == 0.0:
            return den.clamp(-1.0, 1.0)
        flat = den.reshape(b, -1)                                           # components/utils.py:23-33
        scale = torch.quantile(flat.abs(), self.dynamic_threshold, dim=-1).clamp_(min=1.0)
        scale = _extend(scale, den.ndim)
        return den.clamp(-scale, scale) / scale

    # diffusion.py:65-98 (training loss; stock tensor ops, outside the accelerated path)
    def forward(self, x: Tensor, net: nn.Module, sigmas: Tensor, inference: bool = False, cond_scale: float = 1.0,
                **kwargs) -> Tensor:
        if isinstance(net, HipNet) and torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters()):
            raise NotImplementedError("EluDiffusion.forward is the training loss; the HIP UNet1dBase is an inference path without "
                                      "backward -- train the reference module and load its state_dict here, or call under torch.no_grad()")
        noise = torch.randn_like(x)
        x_noisy = x + _extend(sigmas, x.ndim) * noise
        mask = torch.ones_like(x)
        if "x_mask" in kwargs:
            m = kwargs["x_mask"]
            mask = mask * m + torch.ones_like(x) * (~m) * 0.01
        den = self.denoise_fn(x_noisy=x_noisy, net=net, sigmas=sigmas, inference=inference, cond_scale=cond_scale, **kwargs)
        losses = ((den - x) ** 2 * mask).reshape(x.shape[0], -1).sum(dim=1)
        per_sample = float(x[0].numel())
        return losses * self.loss_weight(sigmas) / per_sample
