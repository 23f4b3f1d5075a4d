"""ORACLE (test infrastructure, not product code) -- EDM schedule, preconditioning
and denoise wrapper, restated on CPU fp32.  See oracle/unet1d.py for the rules on who
may import this package and for the parity-pinning status."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from audiodiffuser_amd.config import UNet1dConfig
from .unet1d import unet1d_forward, P


def karras_sigmas(sigma_min: float, sigma_max: float, rho: float, num_steps: int) -> torch.Tensor:
    """src/models/components/scheduler.py:17-22 (EDM eq. 5), fp32."""
    inv = 1.0 / rho
    i = torch.arange(num_steps, dtype=torch.float32)
    return (sigma_max ** inv + i / (num_steps - 1) * (sigma_min ** inv - sigma_max ** inv)) ** rho


def edm_scale_weights(sigmas: torch.Tensor, sigma_data: float, ndim: int) -> Tuple[torch.Tensor, ...]:
    """src/models/components/diffusion.py:232-241 -> (c_skip, c_out, c_in, c_noise)."""
    c_noise = torch.log(sigmas) * 0.25
    s = sigmas.view(*sigmas.shape, *((1,) * (ndim - sigmas.ndim)))   # components/utils.py:16-18
    c_skip = (sigma_data ** 2) / (s ** 2 + sigma_data ** 2)
    c_out = s * sigma_data * (sigma_data ** 2 + s ** 2) ** -0.5
    c_in = (s ** 2 + sigma_data ** 2) ** -0.5
    return c_skip, c_out, c_in, c_noise


def clip(x: torch.Tensor, dynamic_threshold: float = 0.0) -> torch.Tensor:
    """src/models/components/utils.py:19-33: clamp(-1, 1), or dynamic thresholding -- per sample scale = max(1, quantile(|x|, q)),
    x = clamp(x, -scale, scale) / scale."""
    if dynamic_threshold == 0.0:
        return x.clamp(-1.0, 1.0)
    flat = x.reshape(x.shape[0], -1)
    scale = torch.quantile(flat.abs(), dynamic_threshold, dim=-1)
    scale.clamp_(min=1.0)
    scale = scale.view(*scale.shape, *((1,) * (x.ndim - scale.ndim)))
    return x.clamp(-scale, scale) / scale


def denoise(net: Callable[..., torch.Tensor], x_noisy: torch.Tensor,
            sigma_data: float, sigma=None, sigmas: Optional[torch.Tensor] = None, cond_scale: float = 1.0,
            dynamic_threshold: float = 0.0) -> torch.Tensor:
    """src/models/components/diffusion.py:32-63 at inference (clip: components/utils.py:19-33).  Exactly one of sigma / sigmas.
    cond_scale != 1: classifier-free guidance (:52-54); ``net`` must then accept cond_drop_prob."""
    assert (sigma is None) ^ (sigmas is None), "Either sigma or sigmas must be provided"
    b = x_noisy.shape[0]
    if sigmas is None:
        sigmas = torch.full((b,), float(sigma), dtype=torch.float32)   # components/utils.py:41-52
    c_skip, c_out, c_in, c_noise = edm_scale_weights(sigmas, sigma_data, x_noisy.ndim)
    if cond_scale == 1.0:
        pred = net(c_in * x_noisy, c_noise)
    else:
        pred = net(c_in * x_noisy, c_noise, cond_drop_prob=0.0)
        null = net(c_in * x_noisy, c_noise, cond_drop_prob=1.0)
        pred = null + (pred - null) * cond_scale
    return clip(c_skip * x_noisy + c_out * pred, dynamic_threshold)


def make_denoiser(p: P, cfg: UNet1dConfig, sigma_data: float, classes: Optional[torch.Tensor] = None,
                  cond_scale: float = 1.0, storage: str = "fp32", dynamic_threshold: float = 0.0) -> Callable:
    """fn(x, sigma) -> denoised, the closure the samplers call (module call site:
    src/models/diffunet_complex_module.py:86-89); ``classes`` / ``cond_scale`` as the module forwards them.
    ``storage="bf16"``: the network in the bf16-storage arithmetic of oracle/unet1d.py (the preconditioning, the
    clamp and the sampler state stay fp32, as on the device)."""
    def net(xi, t, cond_drop_prob=0.0):
        return unet1d_forward(p, cfg, xi, t, classes=classes, cond_drop_prob=cond_drop_prob, storage=storage)

    def fn(x, sigma=None, sigmas=None):
        return denoise(net, x, sigma_data, sigma=sigma, sigmas=sigmas, cond_scale=cond_scale, dynamic_threshold=dynamic_threshold)
    return fn
