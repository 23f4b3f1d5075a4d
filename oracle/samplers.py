"""ORACLE (test infrastructure, not product code) -- EDM-family sampling loops restated
on CPU fp32.  ``fn(x, sigma=<0-dim fp32 tensor>)`` is the denoiser closure.  All sigma
arithmetic stays on 0-dim fp32 tensors, as in the reference.  See oracle/unet1d.py for
import rules and parity-pinning status."""
from __future__ import annotations

from math import sqrt
from typing import Callable, List, Optional

import torch


def edm_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int,
                s_tmin: float = 0.0, s_tmax: float = float("inf"), s_churn: float = 0.0,
                s_noise: float = 1.0, use_heun: bool = True,
                injected_noise: Optional[torch.Tensor] = None,
                trace: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """src/models/components/sampler_edm.py:371-397 (loop) and :333-369 (step).

    ``injected_noise[i]`` replaces the i-th ``randn_like`` draw (the reference always
    draws, :346, even when gamma == 0)."""
    sig = torch.cat([sigmas, torch.zeros_like(sigmas[:1])])
    x = sig[0] * noise
    gam = torch.where((sig >= s_tmin) & (sig <= s_tmax), min(s_churn / num_steps, sqrt(2) - 1), 0.0)
    for i in range(num_steps):
        s, s_next, g = sig[i], sig[i + 1], gam[i]
        eps = injected_noise[i] if injected_noise is not None else torch.randn_like(x)
        eps = s_noise * eps
        if g > 0:
            s_hat = s + g * s
            x_hat = x + (s_hat ** 2 - s ** 2) ** 0.5 * eps
        else:
            s_hat, x_hat = s, x
        d = (x_hat - fn(x_hat, sigma=s_hat)) / s_hat
        x_next = x_hat + (s_next - s_hat) * d
        if s_next != 0 and use_heun:
            d2 = (x_next - fn(x_next, sigma=s_next)) / s_next
            x_next = x_hat + 0.5 * (s_next - s_hat) * (d + d2)
        x = x_next
        if trace is not None:
            trace.append(x.clone())
    return x


def edm_alpha_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int,
                      alpha: float = 1.0, use_heun: bool = True,
                      trace: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """sampler_edm.py:284-300 (loop, num_steps-1 iterations) and :251-282 (generalised RK2 step)."""
    x = sigmas[0] * noise
    for i in range(num_steps - 1):
        s, s_next = sigmas[i], sigmas[i + 1]
        h = s_next - s
        d = (x - fn(x, sigma=s)) / s
        s_p = s + alpha * h
        if s_p != 0 and use_heun:
            x_p = x + alpha * h * d
            d_p = (x_p - fn(x_p, sigma=s_p)) / s_p
            x = x + h * ((1 - 0.5 / alpha) * d + 0.5 / alpha * d_p)
        else:
            x = x + h * d
        if trace is not None:
            trace.append(x.clone())
    return x


def dpm_multistep_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int,
                          order: int = 3, trace: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """sampler_edm.py:710-768 + :624-690, the shipped setting
    (configs/experiment/sc09_inference/diffunet_complex_sc09_eval_dpm.yaml:57-64):
    multisteps=True, x0_pred=True, log_time_spacing=False.  Then the "lambda" list is the sigma
    list itself (:556), lambd(s) = -log s (:532), the loop makes num_steps-1 updates (:526)
    and ends at sigmas[num_steps-1]; final clamp (:805)."""
    steps = num_steps - 1
    assert steps >= order
    lam = lambda s: -s.log()
    x = sigmas[0] * noise
    s_hist = [sigmas[0]]
    m_hist = [fn(x, sigma=sigmas[0])]

    def update(x, s_cur, ord_):
        s0 = s_hist[-1]
        h = lam(s_cur) - lam(s0)
        phi1 = torch.expm1(-h)
        if ord_ == 1:
            return s_cur / s0 * x - phi1 * m_hist[-1]
        if ord_ == 2:
            h1 = lam(s0) - lam(s_hist[-2])
            r0 = h1 / h
            d1 = (1.0 / r0) * (m_hist[-1] - m_hist[-2])
            return s_cur / s0 * x - phi1 * m_hist[-1] - 0.5 * phi1 * d1
        h1 = lam(s_hist[-2]) - lam(s_hist[-3])
        h0 = lam(s0) - lam(s_hist[-2])
        r0, r1 = h0 / h, h1 / h
        d10 = (1.0 / r0) * (m_hist[-1] - m_hist[-2])
        d11 = (1.0 / r1) * (m_hist[-2] - m_hist[-3])
        d1 = d10 + (r0 / (r0 + r1)) * (d10 - d11)
        d2 = (1.0 / (r0 + r1)) * (d10 - d11)
        phi2 = phi1 / h + 1.0
        phi3 = phi2 / h - 0.5
        return s_cur / s0 * x - phi1 * m_hist[-1] + phi2 * d1 - phi3 * d2

    for step in range(1, steps + 1):
        ord_ = step if step < order else min(order, steps + 1 - step)
        s_cur = sigmas[step]
        x = update(x, s_cur, ord_)
        s_hist.append(s_cur)
        s_hist[:] = s_hist[-order:]
        if step < steps:
            m_hist.append(fn(x, sigma=s_cur))
            m_hist[:] = m_hist[-order:]
        if trace is not None:
            trace.append(x.clone())
    return x.clamp(-1.0, 1.0)


def dpm2_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int,
                 s_tmin: float = 0.0, s_tmax: float = float("inf"), s_churn: float = 150.0, s_noise: float = 1.04,
                 injected_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sampler_edm.py:470-493 (loop, num_steps-1 iterations, final clamp) and :428-468 (step, 'DPM2 Karras').
    Kept as written in the reference: the churned point x_hat only feeds the first derivative, both updates start
    from the UN-churned x (:449, :458, :464); a draw is consumed every step (:439)."""
    x = sigmas[0] * noise
    gam = torch.where((sigmas >= s_tmin) & (sigmas <= s_tmax), min(s_churn / num_steps, sqrt(2) - 1), 0.0)
    for i in range(num_steps - 1):
        s, s_next, g = sigmas[i], sigmas[i + 1], gam[i]
        s_hat = s + g * s
        eps = s_noise * (injected_noise[i] if injected_noise is not None else torch.randn_like(x))
        x_hat = x + (s_hat ** 2 - s ** 2) ** 0.5 * eps if g > 0 else x
        d = (x_hat - fn(x_hat, sigma=s_hat)) / s_hat
        if s_next == 0.0:
            x = x + d * (s_next - s_hat)
        else:
            s_mid = s_hat.log().lerp(s_next.log(), 0.5).exp()
            x_2 = x + d * (s_mid - s_hat)
            d_2 = (x_2 - fn(x_2, sigma=s_mid)) / s_mid
            x = x + d_2 * (s_next - s_hat)
    return x.clamp(-1.0, 1.0)


def adpm2_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int, rho: float = 1.0,
                  eta: float = 1.0, injected_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """stochastic_sampler_edm.py:85-100 (loop, final clamp), :53-83 (step, 'DPM2 a Karras'), :29-32 (get_sigmas)."""
    x = sigmas[0] * noise
    for i in range(num_steps - 1):
        s, s_next = sigmas[i], sigmas[i + 1]
        s_up = min(s_next, eta * (s_next ** 2 * (s ** 2 - s_next ** 2) / s ** 2) ** 0.5)
        s_down = (s_next ** 2 - s_up ** 2) ** 0.5
        d = (x - fn(x, sigma=s)) / s
        s_mid = ((s ** (1 / rho) + s_down ** (1 / rho)) / 2) ** rho
        x_mid = x + d * (s_mid - s)
        d_mid = (x_mid - fn(x_mid, sigma=s_mid)) / s_mid
        x = x + d_mid * (s_down - s)
        x = x + (injected_noise[i] if injected_noise is not None else torch.randn_like(x)) * s_up
    return x.clamp(-1.0, 1.0)
