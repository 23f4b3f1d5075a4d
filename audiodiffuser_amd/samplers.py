"""``model.sampler`` plugins: EDM-family sampling loops
(reference: src/models/components/sampler_edm.py:229-300 ``EDMAlphaSampler``, :302-397 ``EDMSampler``,
:495-805 ``DPMSampler``).

Constructor kwargs and ``forward(noise, fn, net, sigmas, **kwargs)`` follow the reference.  When ``fn``
is ``EluDiffusion.denoise_fn`` of this package and ``net`` is the HIP ``UNet1dBase``, the whole step
loop runs inside ``libadf_hip.so`` (one ``adf_sampler_run`` call, optionally one hipGraph replay): all
branch conditions of the reference loop (``gamma > 0``, ``sigma_next != 0``, warm-up orders) depend only
on the sigma schedule, so they are resolved on the host before anything is enqueued.  For any other
callable the same recurrences are expressed with tensor ops around ``fn`` (interface compatibility).
"""
from __future__ import annotations

import os
from math import sqrt
from typing import Callable, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .diffusion import EluDiffusion
from .net import HipNet, UNet1dBase


# ADF_REQUIRE_NATIVE=1 (or ``samplers.REQUIRE_NATIVE = True``): a sampler call that would NOT run inside libadf_hip.so raises
# instead of taking the interface-compatibility branch.  The GPU test-suite sets it (tests/conftest.py), so a refactor that
# silently sends the package's own (fn, net) pair down the tensor-op restatement fails every sampler parity test.
REQUIRE_NATIVE = os.environ.get("ADF_REQUIRE_NATIVE", "0") not in ("", "0")


def _native_pair(fn: Callable, net, cond_scale: float, kwargs: dict, noise: Optional[Tensor] = None,
                 who: str = "sampler") -> Optional[EluDiffusion]:
    """The whole loop runs inside the HIP library when ``fn`` is this package's ``EluDiffusion.denoise_fn``, ``net`` one of
    its HIP nets and ``noise`` lives on a ROCm device.  The only conditioning it understands is ``classes`` (labels) on a
    class-conditional net, where ``cond_scale != 1`` is classifier-free guidance (two network passes per evaluation).
    Returns the owner of ``fn`` (-> device loop) or None (-> interface-compatibility branch)."""
    owner = getattr(fn, "__self__", None)
    why = None
    if not (isinstance(owner, EluDiffusion) and getattr(fn, "__func__", None) is EluDiffusion.denoise_fn):
        why = "fn is not audiodiffuser_amd.EluDiffusion.denoise_fn"
    elif not isinstance(net, HipNet):
        why = "net is not one of this package's HIP networks"
    elif not 0.0 <= owner.dynamic_threshold <= 1.0:
        why = "dynamic_threshold outside [0, 1]"
    elif noise is not None and not noise.is_cuda:
        why = "noise is not on a ROCm device"
    else:
        extra = {k: v for k, v in kwargs.items() if v is not None}
        if net.cfg.class_cond:
            if set(extra) != {"classes"}:
                why = "a class-conditional net takes exactly the `classes` keyword"
        elif extra or cond_scale != 1.0:
            why = "conditioning keywords / cond_scale != 1 on an unconditional net"
    if why is None:
        return owner
    if REQUIRE_NATIVE:
        raise RuntimeError(f"{who}: ADF_REQUIRE_NATIVE is set and this call would leave the device loop ({why})")
    return None


def _condition(net: UNet1dBase, hd, device, cond_scale: float, kwargs: dict, diff: Optional[EluDiffusion] = None) -> None:
    """Labels + guidance scale of this sampler run (the reference forwards them to every fn call, e.g.
    sampler_edm.py:341-345), and the clipping of the owner of ``fn`` (clamp or dynamic threshold, diffusion.py:61)."""
    hd.set_dynamic_threshold(diff.dynamic_threshold if diff is not None else 0.0)
    if net.cfg.class_cond:
        hd.set_condition(kwargs["classes"], device, null_labels=False, cond_scale=float(cond_scale))


def _draws(x: Tensor, n: int, injected: Optional[Tensor], needed: bool) -> Optional[Tensor]:
    """The per-step ``randn_like`` draws of a reference loop ([n, B, C, L]).  ``injected`` replaces them after a shape
    check (the C side copies n * B * C * L floats from the pointer).  Without it the draws are made here, one
    ``randn_like`` per step in the reference's order -- ALSO when the schedule never uses them (``needed`` False, e.g.
    s_churn = 0): the reference steps draw unconditionally (sampler_edm.py:346, :439), so the global generator is left
    in the state the reference leaves it in and the next batch's initial noise matches seed for seed."""
    if injected is not None:
        if injected.ndim != x.ndim + 1 or tuple(injected.shape[1:]) != tuple(x.shape) or injected.shape[0] < n:
            raise ValueError(f"injected_noise must be shaped [>= {n}, {', '.join(str(d) for d in x.shape)}] (one draw per step), "
                             f"got {tuple(injected.shape)}")
        return injected[:n].detach().to(device=x.device, dtype=torch.float32).contiguous()
    if not needed or n <= 0:
        scratch = torch.empty_like(x)                # the draws only advance the generator: one buffer, n times
        for _ in range(n):
            scratch.normal_()
        return None
    out = torch.empty((n,) + tuple(x.shape), dtype=x.dtype, device=x.device)     # one [n, *x.shape] tensor filled row by row: the same
    for i in range(n):                                                            # generator, the same order as n randn_like calls,
        out[i].normal_()                                                          # half the transient memory of stack()
    return out


def _prep(noise: Tensor) -> Tensor:
    if noise.ndim not in (3, 4):
        raise ValueError("the HIP sampler expects waveforms shaped [B, C, L] (or [B, C, H, W] for the 2-D UNetModel)")
    return noise.detach().to(torch.float32).contiguous()


class EDMSampler(nn.Module):
    """EDM stochastic sampler (Heun, optional churn); ``s_churn=0`` is the deterministic Heun ODE solver."""

    def __init__(self, s_tmin: float = 0, s_tmax: float = float("inf"), s_churn: float = 150.0, s_noise: float = 1.04,
                 num_steps: int = 200, cond_scale: float = 1.0, use_heun: bool = True, use_graph: bool = True):
        super().__init__()
        self.s_tmin, self.s_tmax, self.s_noise, self.s_churn = s_tmin, s_tmax, s_noise, s_churn
        self.num_steps, self.cond_scale, self.use_heun, self.use_graph = num_steps, cond_scale, use_heun, use_graph

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_EDM, self.num_steps
        d.s_tmin, d.s_tmax, d.s_churn, d.s_noise = self.s_tmin, min(self.s_tmax, 3.0e38), self.s_churn, self.s_noise
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = int(self.use_heun), 1.0, 0, sigma_data, int(self.use_graph)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, injected_noise: Optional[Tensor] = None,
                **kwargs) -> Tensor:
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            inj = _draws(x, self.num_steps, injected_noise, self.s_churn > 0)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, inj).to(noise.dtype)
        # ---- interface-compatibility branch (sampler_edm.py:333-397) -----------------------------
        sig = torch.cat([sigmas, torch.zeros_like(sigmas[:1])])
        x = sig[0] * noise
        gam = torch.where((sig >= self.s_tmin) & (sig <= self.s_tmax), min(self.s_churn / self.num_steps, sqrt(2) - 1), 0.0)
        for i in range(self.num_steps):
            s, s_next, g = sig[i], sig[i + 1], gam[i]
            eps = injected_noise[i] if injected_noise is not None else torch.randn_like(x)
            eps = self.s_noise * eps
            if g > 0:
                s_hat = s + g * s
                x_hat = x + (s_hat ** 2 - s ** 2) ** 0.5 * eps
            else:
                s_hat, x_hat = s, x
            d = (x_hat - fn(x_hat, net=net, sigma=s_hat, inference=True, cond_scale=self.cond_scale, **kwargs)) / s_hat
            x = x_hat + (s_next - s_hat) * d
            if s_next != 0 and self.use_heun:
                d2 = (x - fn(x, net=net, sigma=s_next, inference=True, cond_scale=self.cond_scale, **kwargs)) / s_next
                x = x_hat + 0.5 * (s_next - s_hat) * (d + d2)
        return x


class EDMAlphaSampler(nn.Module):
    """EDM algorithm 3, generalised second-order Runge-Kutta; ``alpha=1`` is Heun."""

    def __init__(self, alpha: float = 1.0, num_steps: int = 50, cond_scale: float = 1.0, use_heun: bool = True,
                 use_graph: bool = True):
        super().__init__()
        self.alpha, self.num_steps, self.cond_scale, self.use_heun, self.use_graph = alpha, num_steps, cond_scale, use_heun, use_graph

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_EDM_ALPHA, self.num_steps
        d.s_tmin = d.s_tmax = d.s_churn = 0.0
        d.s_noise = 1.0
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = int(self.use_heun), self.alpha, 0, sigma_data, int(self.use_graph)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, **kwargs) -> Tensor:
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, None).to(noise.dtype)
        x = sigmas[0] * noise                                            # sampler_edm.py:284-300
        for i in range(self.num_steps - 1):
            s, s_next = sigmas[i], sigmas[i + 1]
            h = s_next - s
            d = (x - fn(x, net=net, sigma=s, inference=True, cond_scale=self.cond_scale, **kwargs)) / s
            s_p = s + self.alpha * h
            if s_p != 0 and self.use_heun:
                x_p = x + self.alpha * h * d
                d_p = (x_p - fn(x_p, net=net, sigma=s_p, inference=True, cond_scale=self.cond_scale, **kwargs)) / s_p
                x = x + h * ((1 - 0.5 / self.alpha) * d + 0.5 / self.alpha * d_p)
            else:
                x = x + h * d
        return x


class DPMSampler(nn.Module):
    """DPM-Solver with x0 prediction (sampler_edm.py:495-805): the multistep solver (``multisteps=True``; the shipped
    configuration is configs/experiment/sc09_inference/diffunet_complex_sc09_eval_dpm.yaml:57-64 with
    ``log_time_spacing=False``) and the single-step "DPM-Solver-fast" (``multisteps=False``), each on the sigma grid
    itself or on a grid linear in log sigma.  The reference's quirks are kept: without log spacing the single-step run
    walks only ``len(orders)`` intervals of the sigma list (stops early) and its intermediate points add a
    lambda-space step to a sigma (:584, :604).  ``x0_pred=False`` is the noise-prediction form of every update (:700-706); combined with the single-step
    solver on the sigma grid the reference itself returns NaN, and so does this class."""

    def __init__(self, cond_scale, order=1, num_steps=10, multisteps=False, x0_pred: bool = True,
                 log_time_spacing: bool = True, use_graph: bool = True):
        super().__init__()
        self.order, self.cond_scale, self.multisteps = order, cond_scale, multisteps
        self.x0_pred, self.log_time_spacing, self.use_graph = x0_pred, log_time_spacing, use_graph
        self.ctor_num_steps = num_steps
        self.num_steps = num_steps if log_time_spacing else num_steps - 1      # sampler_edm.py:526

    def _check_supported(self) -> None:
        if self.order not in (1, 2, 3):
            raise ValueError("'order' must be '1' or '2' or '3'.")

    def singlestep_orders(self):
        """sampler_edm.py:770-789."""
        n, order = self.num_steps, self.order
        if order == 3:
            k = n // 3 + 1
            return ([3] * (k - 2) + [2, 1] if n % 3 == 0 else [3] * (k - 1) + [n % 3]), k
        if order == 2:
            return ([2] * (n // 2) if n % 2 == 0 else [2] * (n // 2) + [1]), (n + 1) // 2
        return [1] * n, n

    def nfe(self) -> int:
        return self.num_steps if self.multisteps else sum(self.singlestep_orders()[0])

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind = _lib.SAMPLER_DPM_MULTISTEP if self.multisteps else _lib.SAMPLER_DPM_SINGLESTEP
        d.num_steps = self.ctor_num_steps
        d.s_tmin = d.s_tmax = d.s_churn = 0.0
        d.s_noise = 1.0
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = 0, 1.0, int(self.order), sigma_data, int(self.use_graph)
        d.log_time_spacing = int(bool(self.log_time_spacing))
        d.eps_pred = int(not self.x0_pred)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, **kwargs) -> Tensor:
        self._check_supported()
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, None).to(noise.dtype)
        # ---- interface-compatibility branch (sampler_edm.py:710-805, :568-690) --------------------
        if not self.x0_pred:
            raise NotImplementedError("DPMSampler(x0_pred=False) with a foreign net / fn: noise prediction runs on the native path only")
        call = lambda x, s: fn(x, net=net, sigma=s, inference=True, cond_scale=self.cond_scale, **kwargs)
        if self.log_time_spacing:           # grid of lambda = -log sigma (:546-552); lambd = inv_lambd = identity
            lam = inv = lambda v: v
            sig = lambda l: l.neg().exp()
            grid_of = lambda n: torch.linspace(-sigmas[0].log(), -sigmas[-1].log(), n + 1)
        else:                               # the grid is the sigma list (:556)
            lam = lambda v: -v.log()
            sig = lambda v: v
            inv = lambda l: l.neg().exp()
            grid_of = lambda n: sigmas
        x = sigmas[0] * noise
        if not self.multisteps:
            orders, k = self.singlestep_orders()
            grid = grid_of(k)
            for i, o in enumerate(orders):
                cur, nxt = grid[i], grid[i + 1]
                h = lam(nxt) - lam(cur)
                eps = call(x, sig(cur))
                base = sig(nxt) / sig(cur) * x - torch.expm1(-h) * eps
                if o == 1:
                    x = base
                elif o == 2:
                    r1 = 1 / 2
                    s1 = inv(cur + r1 * h)
                    u1 = sig(s1) / sig(cur) * x - torch.expm1(-r1 * h) * eps
                    x = base - 1 / (2 * r1) * torch.expm1(-h) * (call(u1, sig(s1)) - eps)
                else:
                    r1, r2 = 1 / 3, 2 / 3
                    s1, s2 = inv(cur + r1 * h), inv(cur + r2 * h)
                    u1 = sig(s1) / sig(cur) * x - (-r1 * h).expm1() * eps
                    eps_r1 = call(u1, sig(s1))
                    u2 = (sig(s2) / sig(cur) * x - (-r2 * h).expm1() * eps
                          + (r2 / r1) * ((-r2 * h).expm1() / (r2 * h) + 1) * (eps_r1 - eps))
                    x = base + 1 / r2 * (torch.expm1(-h) / h + 1) * (call(u2, sig(s2)) - eps)
            return x.clamp(-1.0, 1.0)
        steps, order = self.num_steps, self.order
        assert steps >= order
        grid = grid_of(steps)
        s_hist, m_hist = [grid[0]], [call(x, sig(grid[0]))]
        for step in range(1, steps + 1):
            o = step if step < order else min(order, steps + 1 - step)
            s_cur, s0 = grid[step], s_hist[-1]
            h = lam(s_cur) - lam(s0)
            phi1 = torch.expm1(-h)
            new = sig(s_cur) / sig(s0) * x - phi1 * m_hist[-1]
            if o == 2:
                r0 = (lam(s0) - lam(s_hist[-2])) / h
                new = new - 0.5 * phi1 * ((1.0 / r0) * (m_hist[-1] - m_hist[-2]))
            elif o == 3:
                r0 = (lam(s0) - lam(s_hist[-2])) / h
                r1 = (lam(s_hist[-2]) - lam(s_hist[-3])) / h
                d10 = (1.0 / r0) * (m_hist[-1] - m_hist[-2])
                d11 = (1.0 / r1) * (m_hist[-2] - m_hist[-3])
                d1 = d10 + (r0 / (r0 + r1)) * (d10 - d11)
                d2 = (1.0 / (r0 + r1)) * (d10 - d11)
                phi2 = phi1 / h + 1.0
                phi3 = phi2 / h - 0.5
                new = new + phi2 * d1 - phi3 * d2
            x = new
            s_hist = (s_hist + [s_cur])[-order:]
            if step < steps:
                m_hist = (m_hist + [call(x, sig(s_cur))])[-order:]
        return x.clamp(-1.0, 1.0)


class DPM2MSampler(nn.Module):
    """'DPM-Solver++(2M) Karras' (sampler_edm.py:1056-1131).  The loop reads ``sigmas[i + 1]`` for ``i < num_steps``: the schedule
    must hold ``num_steps + 1`` entries (a final 0 returns the last denoised estimate); with the module's own N-entry schedule the
    reference raises IndexError on its last step, and so does this class.  ``reflow`` is the constructor flag of the class of the same
    name in stochastic_sampler_edm.py:180-259 (otherwise the same recurrence): the network output is read as a velocity,
    ``denoised = x - output * sigma`` (:214-215)."""

    def __init__(self, num_steps: int = 50, cond_scale: float = 1.0, reflow: bool = False, use_graph: bool = True):
        super().__init__()
        self.num_steps, self.cond_scale, self.reflow, self.use_graph = num_steps, cond_scale, reflow, use_graph

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_DPM2M, int(self.num_steps)
        d.s_tmin = d.s_tmax = d.s_churn = 0.0
        d.s_noise = 1.0
        d.reflow = int(bool(self.reflow))
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = 0, 1.0, 2, sigma_data, int(self.use_graph)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, **kwargs) -> Tensor:
        if len(sigmas) < self.num_steps + 1:
            raise IndexError(f"index {self.num_steps} is out of bounds for dimension 0 with size {len(sigmas)}")
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, None).to(noise.dtype)
        # ---- interface-compatibility branch --------------------------------------------------------
        x = sigmas[0] * noise
        old = None
        for i in range(self.num_steps):
            s_last, s, s_next = sigmas[i - 1], sigmas[i], sigmas[i + 1]
            den = fn(x, net=net, sigma=s, inference=True, cond_scale=self.cond_scale, **kwargs)
            if self.reflow:
                den = x - den * s
            t, t_next = s.log().neg(), s_next.log().neg()
            h = t_next - t
            t_min, t_max = min(t_next.neg().exp(), t.neg().exp()), max(t_next.neg().exp(), t.neg().exp())
            if old is None or s_next == 0:
                x = (t_min / t_max) * x - (-h).expm1() * den
            else:
                h_last = t - s_last.log().neg()
                h_min, h_max = min(h_last, h), max(h_last, h)
                r = h_max / h_min
                h_d = (h_max + h_min) / 2
                x = (t_min / t_max) * x - (-h_d).expm1() * ((1 + 1 / (2 * r)) * den - (1 / (2 * r)) * old)
            old = den
        return x.clamp(-1.0, 1.0)


class LMSSampler(nn.Module):
    """'LMS Karras' linear multistep solver (sampler_edm.py:1134-1190): ``num_steps - 1`` evaluations, the last ``order``
    derivatives combined with the integrals of their Lagrange basis polynomials over the step (the reference integrates
    them with scipy ``quad`` on fp32 NumPy scalars, so its values carry ~1e-7 relative noise that depends on the NumPy
    version's promotion rules; they are cubics at most, so 3-point Gauss-Legendre in double is exact)."""

    def __init__(self, num_steps: int = 50, cond_scale: float = 1.0, order: int = 4, use_graph: bool = True):
        super().__init__()
        self.num_steps, self.cond_scale, self.order, self.use_graph = num_steps, cond_scale, order, use_graph

    @staticmethod
    def linear_multistep_coeff(order: int, t, i: int, j: int) -> float:
        if order - 1 > i:
            raise ValueError(f"Order {order} too high for step {i}")
        a, b = float(t[i]), float(t[i + 1])
        half, mid, acc = 0.5 * (b - a), 0.5 * (a + b), 0.0
        for gx, gw in ((-0.7745966692414834, 5 / 9), (0.0, 8 / 9), (0.7745966692414834, 5 / 9)):
            tau, prod = mid + half * gx, 1.0
            for k in range(order):
                if k != j:
                    prod *= (tau - float(t[i - k])) / (float(t[i - j]) - float(t[i - k]))
            acc += gw * prod
        return acc * half

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_LMS, int(self.num_steps)
        d.s_tmin = d.s_tmax = d.s_churn = 0.0
        d.s_noise = 1.0
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = 0, 1.0, int(self.order), sigma_data, int(self.use_graph)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, **kwargs) -> Tensor:
        if not 1 <= self.order <= 4:
            raise ValueError("LMSSampler: order must be 1..4")
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, None).to(noise.dtype)
        # ---- interface-compatibility branch --------------------------------------------------------
        t = sigmas.detach().cpu().numpy()
        x = sigmas[0] * noise
        ds = []
        for i in range(self.num_steps - 1):
            d = (x - fn(x, net=net, sigma=sigmas[i], inference=True, cond_scale=self.cond_scale, **kwargs)) / sigmas[i]
            ds = (ds + [d])[-self.order:]
            cur = min(i + 1, self.order)
            coeffs = [self.linear_multistep_coeff(cur, t, i, j) for j in range(cur)]
            x = x + sum(c * dd for c, dd in zip(coeffs, reversed(ds)))
        return x.clamp(-1.0, 1.0)


class DPM2Sampler(nn.Module):
    """'DPM2 Karras' (reference: src/models/components/sampler_edm.py:401-493): midpoint method in log-sigma with the
    EDM churn.  ``injected_noise`` ([num_steps-1, B, C, L]) replaces the per-step ``randn_like`` draws."""

    def __init__(self, rho: float = 2.0, num_steps: int = 50, cond_scale: float = 1.0, s_tmin: float = 0,
                 s_tmax: float = float("inf"), s_churn: float = 150.0, s_noise: float = 1.04, use_graph: bool = True):
        super().__init__()
        self.rho, self.num_steps, self.cond_scale = rho, num_steps, cond_scale
        self.s_tmin, self.s_tmax, self.s_noise, self.s_churn, self.use_graph = s_tmin, s_tmax, s_noise, s_churn, use_graph

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_DPM2, self.num_steps
        d.s_tmin, d.s_tmax, d.s_churn, d.s_noise = self.s_tmin, min(self.s_tmax, 3.0e38), self.s_churn, self.s_noise
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = 0, 1.0, 0, sigma_data, int(self.use_graph)
        d.rho, d.eta = float(self.rho), 1.0
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, injected_noise: Optional[Tensor] = None,
                **kwargs) -> Tensor:
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            inj = _draws(x, self.num_steps - 1, injected_noise, self.s_churn > 0)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, inj).to(noise.dtype)
        # ---- interface-compatibility branch (sampler_edm.py:428-493) ----------------------------------
        x = sigmas[0] * noise
        gam = torch.where((sigmas >= self.s_tmin) & (sigmas <= self.s_tmax), min(self.s_churn / self.num_steps, sqrt(2) - 1), 0.0)
        for i in range(self.num_steps - 1):
            s, s_next, g = sigmas[i], sigmas[i + 1], gam[i]
            s_hat = s + g * s
            eps = self.s_noise * (injected_noise[i] if injected_noise is not None else torch.randn_like(x))
            x_hat = x + (s_hat ** 2 - s ** 2) ** 0.5 * eps if g > 0 else x
            d = (x_hat - fn(x_hat, net=net, sigma=s_hat, inference=True, cond_scale=self.cond_scale, **kwargs)) / s_hat
            if s_next == 0.0:
                x = x + d * (s_next - s_hat)
            else:
                s_mid = s_hat.log().lerp(s_next.log(), 0.5).exp()
                x_2 = x + d * (s_mid - s_hat)
                d_2 = (x_2 - fn(x_2, net=net, sigma=s_mid, inference=True, cond_scale=self.cond_scale, **kwargs)) / s_mid
                x = x + d_2 * (s_next - s_hat)
        return x.clamp(-1.0, 1.0)


class ADPM2Sampler(nn.Module):
    """'DPM2 a Karras', the ancestral DPM-Solver-2 (reference: src/models/components/stochastic_sampler_edm.py:35-100;
    the Lightning module's default sampler).  Fresh noise of scale sigma_up is added every step:
    ``injected_noise`` ([num_steps-1, B, C, L]) replaces those ``randn_like`` draws."""

    def __init__(self, rho: float = 1.0, num_steps: int = 50, cond_scale: float = 1.0, eta: float = 1.0, use_graph: bool = True):
        super().__init__()
        self.rho, self.num_steps, self.cond_scale, self.eta, self.use_graph = rho, num_steps, cond_scale, eta, use_graph

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_ADPM2, self.num_steps
        d.s_tmin, d.s_tmax, d.s_churn, d.s_noise = 0.0, 3.0e38, 0.0, 1.0
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = 0, 1.0, 0, sigma_data, int(self.use_graph)
        d.rho, d.eta = float(self.rho), float(self.eta)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, injected_noise: Optional[Tensor] = None,
                **kwargs) -> Tensor:
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            inj = _draws(x, self.num_steps - 1, injected_noise, True)                          # reference draw order (:82)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, inj).to(noise.dtype)
        # ---- interface-compatibility branch (stochastic_sampler_edm.py:29-32, :53-100) -----------------
        x = sigmas[0] * noise
        for i in range(self.num_steps - 1):
            s, s_next = sigmas[i], sigmas[i + 1]
            s_up = min(s_next, self.eta * (s_next ** 2 * (s ** 2 - s_next ** 2) / s ** 2) ** 0.5)
            s_down = (s_next ** 2 - s_up ** 2) ** 0.5
            d = (x - fn(x, net=net, sigma=s, inference=True, cond_scale=self.cond_scale, **kwargs)) / s
            s_mid = ((s ** (1 / self.rho) + s_down ** (1 / self.rho)) / 2) ** self.rho
            x_mid = x + d * (s_mid - s)
            d_mid = (x_mid - fn(x_mid, net=net, sigma=s_mid, inference=True, cond_scale=self.cond_scale, **kwargs)) / s_mid
            x = x + d_mid * (s_down - s)
            x = x + (injected_noise[i] if injected_noise is not None else torch.randn_like(x)) * s_up
        return x.clamp(-1.0, 1.0)


class ADPMPP2SSampler(nn.Module):
    """'DPM++ 2S a Karras', ancestral DPM-Solver++(2S) (reference: src/models/components/stochastic_sampler_edm.py:102-178):
    ``num_steps - 1`` steps of two evaluations (one when sigma_down is 0), fresh noise of scale sigma_up after every step whose
    sigma_next is positive -- ``injected_noise`` ([that many, B, C, L]) replaces those ``randn_like`` draws.  ``rho`` is accepted and,
    as in the reference, never read."""

    def __init__(self, rho: float = 1.0, num_steps: int = 50, cond_scale: float = 1.0, eta: float = 1.0, use_graph: bool = True):
        super().__init__()
        self.rho, self.num_steps, self.cond_scale, self.eta, self.use_graph = rho, num_steps, cond_scale, eta, use_graph

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_ADPMPP2S, self.num_steps
        d.s_tmin, d.s_tmax, d.s_churn, d.s_noise = 0.0, 3.0e38, 0.0, 1.0
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = 0, 1.0, 0, sigma_data, int(self.use_graph)
        d.rho, d.eta = float(self.rho), float(self.eta)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, injected_noise: Optional[Tensor] = None,
                **kwargs) -> Tensor:
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            n_draws = int((sigmas[1:self.num_steps].detach().cpu() > 0).sum())                 # one per step with sigma_next > 0 (:158)
            inj = _draws(x, n_draws, injected_noise, True)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, inj).to(noise.dtype)
        # ---- interface-compatibility branch (stochastic_sampler_edm.py:29-32, :117-178) ----------------
        x = sigmas[0] * noise
        k = 0
        for i in range(self.num_steps - 1):
            s, s_next = sigmas[i], sigmas[i + 1]
            den = fn(x, net=net, sigma=s, inference=True, cond_scale=self.cond_scale, **kwargs)
            s_up = min(s_next, self.eta * (s_next ** 2 * (s ** 2 - s_next ** 2) / s ** 2) ** 0.5)
            s_down = (s_next ** 2 - s_up ** 2) ** 0.5
            if s_down == 0:
                x = x + (x - den) / s * (s_down - s)
            else:
                t, t_next = s.log().neg(), s_down.log().neg()
                h = t_next - t
                sm = t + 0.5 * h
                x_2 = (sm.neg().exp() / t.neg().exp()) * x - (-h * 0.5).expm1() * den
                den_2 = fn(x_2, net=net, sigma=sm.neg().exp(), inference=True, cond_scale=self.cond_scale, **kwargs)
                x = (t_next.neg().exp() / t.neg().exp()) * x - (-h).expm1() * den_2
            if s_next > 0:
                x = x + (injected_noise[k] if injected_noise is not None else torch.randn_like(x)) * s_up
                k += 1
        return x.clamp(-1.0, 1.0)


class UniPCSampler(nn.Module):
    """Uni-PC (sampler_edm.py:807-1053, variant 'bh2'): multistep predictor-corrector, NFE = its step count -- ``num_steps`` on a lambda
    grid linear between the first and the last sigma (``log_time_spacing``, the default) or ``num_steps - 1`` on the sigma list itself.
    The reference only runs on 4-D states (hard-coded ``einsum('k,bkchw->bchw')``); here any state shape works."""

    def __init__(self, num_steps: int = 20, order: int = 2, cond_scale: float = 1.0, x0_pred: bool = True,
                 log_time_spacing: bool = True, use_graph: bool = True):
        super().__init__()
        self.order, self.cond_scale, self.x0_pred, self.log_time_spacing = order, cond_scale, x0_pred, log_time_spacing
        self._ctor_steps = num_steps
        self.num_steps = num_steps if log_time_spacing else num_steps - 1        # as the reference stores it (:828)
        self.use_graph = use_graph

    def _desc(self, sigma_data: float) -> "_lib.AdfSamplerDesc":
        d = _lib.AdfSamplerDesc()
        d.kind, d.num_steps = _lib.SAMPLER_UNIPC, int(self._ctor_steps)
        d.s_tmin = d.s_tmax = d.s_churn = 0.0
        d.s_noise = 1.0
        d.use_heun, d.alpha, d.order, d.sigma_data, d.use_graph = 0, 1.0, int(self.order), sigma_data, int(self.use_graph)
        d.log_time_spacing, d.eps_pred = int(self.log_time_spacing), int(not self.x0_pred)
        return d

    @torch.no_grad()
    def forward(self, noise: Tensor, fn: Callable, net: nn.Module, sigmas: Tensor, **kwargs) -> Tensor:
        assert self.num_steps >= self.order                                        # :1001
        diff = _native_pair(fn, net, self.cond_scale, kwargs, noise, type(self).__name__)
        if diff is not None:
            x = _prep(noise)
            hd = net.native(x.device)
            _condition(net, hd, x.device, self.cond_scale, kwargs, diff)
            return hd.sampler_run(self._desc(diff.sigma_data), sigmas, x, None).to(noise.dtype)
        # ---- interface-compatibility branch (same recurrences as tensor ops around a foreign fn / net) ----------------------------
        steps, order = self.num_steps, self.order
        if self.log_time_spacing:
            grid = torch.linspace(-sigmas[0].log(), -sigmas[-1].log(), steps + 1).to(noise.device)
            lam, sig = (lambda v: v), (lambda v: v.neg().exp())
        else:
            grid = sigmas.to(noise.device)
            lam, sig = (lambda v: -v.log()), (lambda v: v)

        def model(x_, g):
            den = fn(x_, net=net, sigma=sig(g), inference=True, cond_scale=self.cond_scale, **kwargs)
            return den if self.x0_pred else (x_ - den) / sig(g)

        def update(x_, ml, gl, g_cur, ord_, corr):
            g0, m0 = gl[-1], ml[-1]
            h = (lam(g_cur) - lam(g0)).view(-1)
            rks, d1s = [], []
            for i in range(1, ord_):
                rk = (lam(gl[-(i + 1)]) - lam(g0)) / h
                rks.append(rk)
                d1s.append((ml[-(i + 1)] - m0) / rk)
            rks = torch.tensor([float(r) for r in rks] + [1.0], device=noise.device)
            hh = -h if self.x0_pred else h
            h_phi_1 = torch.expm1(hh)
            h_phi_k, b_h, fact = h_phi_1 / hh - 1, torch.expm1(hh), 1
            R, b = [], []
            for i in range(1, ord_ + 1):
                R.append(torch.pow(rks, i - 1))
                b.append(h_phi_k * fact / b_h)
                fact *= i + 1
                h_phi_k = h_phi_k / hh - 1 / fact
            R, b = torch.stack(R), torch.cat(b)
            rho_p = (torch.tensor([0.5], device=noise.device) if ord_ == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])) if d1s else None
            scale = 1.0 if self.x0_pred else sig(g_cur)
            xt_ = sig(g_cur) / sig(g0) * x_ - h_phi_1 * m0 if self.x0_pred else x_ - sig(g_cur) * h_phi_1 * m0
            comb = lambda rho: sum(rho[k] * d for k, d in enumerate(d1s)) if d1s else 0
            x_t, m_t = xt_ - scale * b_h * comb(rho_p), None
            if corr:
                rho_c = torch.tensor([0.5], device=noise.device) if ord_ == 1 else torch.linalg.solve(R, b)
                m_t = model(x_t, g_cur)
                x_t = xt_ - scale * b_h * (comb(rho_c[:-1]) + rho_c[-1] * (m_t - m0))
            return x_t, m_t

        x = sigmas[0] * noise
        ml, gl = [model(x, grid[0])], [grid[0]]
        for step in range(1, order):
            x, m = update(x, ml, gl, grid[step], step, True)
            gl.append(grid[step]); ml.append(m)
        for step in range(order, steps + 1):
            x, m = update(x, ml, gl, grid[step], min(order, steps + 1 - step), step != steps)
            for i in range(order - 1):
                gl[i], ml[i] = gl[i + 1], ml[i + 1]
            gl[-1] = grid[step]
            if step < steps:
                ml[-1] = m
        return x.clamp(-1.0, 1.0)
