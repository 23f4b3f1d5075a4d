"""``model.noise_scheduler`` plugin (reference: src/models/components/scheduler.py:6-22).

The Lightning module stores ``noise_scheduler()`` once (diffunet_complex_module.py:64), i.e. the
instance call returns the fp32 sigma tensor on the CPU; it is moved to the device at the call site.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor


class KarrasSchedule(nn.Module):
    """EDM eq. 5: sigma_i = (smax^(1/rho) + i/(N-1) (smin^(1/rho) - smax^(1/rho)))^rho."""

    def __init__(self, sigma_min: float, sigma_max: float, rho: float = 7.0, num_steps: int = 50):
        super().__init__()
        self.sigma_min, self.sigma_max, self.rho, self.num_steps = sigma_min, sigma_max, rho, num_steps

    def forward(self) -> Tensor:
        inv = 1.0 / self.rho
        ramp = torch.arange(self.num_steps, dtype=torch.float32) / (self.num_steps - 1)
        lo, hi = self.sigma_min ** inv, self.sigma_max ** inv
        return (hi + ramp * (lo - hi)) ** self.rho
