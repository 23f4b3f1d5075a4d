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


def _dpm_grid(sigmas: torch.Tensor, n_intervals: int, log_time_spacing: bool):
    """DPMSampler.get_lambda / lambd / sigma / inv_lambd (sampler_edm.py:528-556).  Returns (grid, lam, sig, inv):
    log spacing: the grid holds lambda = -log sigma, linspace over n_intervals + 1 points between the first and the
    last sigma, lam = inv = identity, sig(l) = exp(-l); otherwise the grid IS the sigma list, lam(s) = -log s,
    sig = identity, inv(l) = exp(-l)."""
    if log_time_spacing:
        grid = torch.linspace(-sigmas[0].log(), -sigmas[-1].log(), n_intervals + 1)
        ident = lambda v: v
        return grid, ident, (lambda l: l.neg().exp()), ident
    return sigmas, (lambda s_: -s_.log()), (lambda v: v), (lambda l: l.neg().exp())


def dpm_multistep_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int,
                          order: int = 3, trace: Optional[List[torch.Tensor]] = None,
                          log_time_spacing: bool = False, x0_pred: bool = True) -> torch.Tensor:
    """sampler_edm.py:710-768 + :624-690 with multisteps=True, x0_pred=True.  The shipped setting
    (configs/experiment/sc09_inference/diffunet_complex_sc09_eval_dpm.yaml:57-64) has log_time_spacing=False: the
    "lambda" list is the sigma list itself (:556), lambd(s) = -log s (:532), the loop makes num_steps-1 updates (:526)
    and ends at sigmas[num_steps-1].  log_time_spacing=True: num_steps updates on a lambda grid that is linear between
    the first and the last sigma (:549-552).  Final clamp (:805)."""
    steps = num_steps if log_time_spacing else num_steps - 1
    assert steps >= order
    grid, lam, sig, _ = _dpm_grid(sigmas, steps, log_time_spacing)
    x = sigmas[0] * noise
    # x0_pred=False: the model value is the noise prediction (x - D(x)) / sigma (:700-706) and the updates take their
    # noise-prediction form (:640-645, :660-662, :685-689)
    model = (lambda x_, g: fn(x_, sigma=sig(g))) if x0_pred else (lambda x_, g: (x_ - fn(x_, sigma=sig(g))) / sig(g))
    s_hist = [grid[0]]
    m_hist = [model(x, grid[0])]

    def update(x, s_cur, ord_):
        s0 = s_hist[-1]
        h = lam(s_cur) - lam(s0)
        if not x0_pred:
            phi1 = torch.expm1(h)
            if ord_ == 1:
                return x - sig(s_cur) * phi1 * m_hist[-1]
            if ord_ == 2:
                r0 = (lam(s0) - lam(s_hist[-2])) / h
                d1 = (1.0 / r0) * (m_hist[-1] - m_hist[-2])
                return x - (sig(s_cur) * phi1) * m_hist[-1] - 0.5 * (sig(s_cur) * phi1) * d1
            h1 = lam(s_hist[-2]) - lam(s_hist[-3])
            h0 = lam(s0) - lam(s_hist[-2])
            r0, r1 = h0 / h, h1 / h
            d10 = (1.0 / r0) * (m_hist[-1] - m_hist[-2])
            d11 = (1.0 / r1) * (m_hist[-2] - m_hist[-3])
            d1 = d10 + (r0 / (r0 + r1)) * (d10 - d11)
            d2 = (1.0 / (r0 + r1)) * (d10 - d11)
            phi2 = phi1 / h - 1.0
            phi3 = phi2 / h - 0.5
            return x - (sig(s_cur) * phi1) * m_hist[-1] - (sig(s_cur) * phi2) * d1 - (sig(s_cur) * phi3) * d2
        phi1 = torch.expm1(-h)
        if ord_ == 1:
            return sig(s_cur) / sig(s0) * x - phi1 * m_hist[-1]
        if ord_ == 2:
            h1 = lam(s0) - lam(s_hist[-2])
            r0 = h1 / h
            d1 = (1.0 / r0) * (m_hist[-1] - m_hist[-2])
            return sig(s_cur) / sig(s0) * x - phi1 * m_hist[-1] - 0.5 * phi1 * d1
        h1 = lam(s_hist[-2]) - lam(s_hist[-3])
        h0 = lam(s0) - lam(s_hist[-2])
        r0, r1 = h0 / h, h1 / h
        d10 = (1.0 / r0) * (m_hist[-1] - m_hist[-2])
        d11 = (1.0 / r1) * (m_hist[-2] - m_hist[-3])
        d1 = d10 + (r0 / (r0 + r1)) * (d10 - d11)
        d2 = (1.0 / (r0 + r1)) * (d10 - d11)
        phi2 = phi1 / h + 1.0
        phi3 = phi2 / h - 0.5
        return sig(s_cur) / sig(s0) * x - phi1 * m_hist[-1] + phi2 * d1 - phi3 * d2

    for step in range(1, steps + 1):
        ord_ = step if step < order else min(order, steps + 1 - step)
        s_cur = grid[step]
        x = update(x, s_cur, ord_)
        s_hist.append(s_cur)
        s_hist[:] = s_hist[-order:]
        if step < steps:
            m_hist.append(model(x, s_cur))
            m_hist[:] = m_hist[-order:]
        if trace is not None:
            trace.append(x.clone())
    return x.clamp(-1.0, 1.0)


def dpm_singlestep_orders(num_steps_eff: int, order: int) -> List[int]:
    """sampler_edm.py:770-789: the orders of the single-step solver ("DPM-Solver-fast")."""
    n = num_steps_eff
    if order == 3:
        k = n // 3 + 1
        return [3] * (k - 2) + [2, 1] if n % 3 == 0 else [3] * (k - 1) + [n % 3]
    if order == 2:
        return [2] * (n // 2) if n % 2 == 0 else [2] * (n // 2) + [1]
    if order == 1:
        return [1] * n
    raise ValueError("'order' must be '1' or '2' or '3'.")


def dpm_singlestep_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int, order: int = 3,
                           log_time_spacing: bool = True, x0_pred: bool = True) -> torch.Tensor:
    """sampler_edm.py:769-805 (the multisteps=False branch) + :568-622 (dpm_solver_{1,2,3}_step), x0_pred=True.
    Kept as written: (a) with log_time_spacing=False the grid is the full sigma list but only len(orders) intervals are
    walked, so the run stops early (:791-795); (b) in that mode the intermediate points add a lambda-space step to a
    SIGMA (`s1 = lambd_cur + r1 * h`, :584/:604) before inv_lambd; (c) the first evaluation of every interval is made
    once and cached (:796).  num_steps_eff = num_steps (log spacing) or num_steps - 1 (:526)."""
    n_eff = num_steps if log_time_spacing else num_steps - 1
    orders = dpm_singlestep_orders(n_eff, order)
    k = {3: n_eff // 3 + 1, 2: (n_eff + 1) // 2, 1: n_eff}[order]
    grid, lam, sig, inv = _dpm_grid(sigmas, k, log_time_spacing)
    x = sigmas[0] * noise
    model = (lambda x_, g: fn(x_, sigma=sig(g))) if x0_pred else (lambda x_, g: (x_ - fn(x_, sigma=sig(g))) / sig(g))
    for i, o in enumerate(orders):
        cur, nxt = grid[i], grid[i + 1]
        h = lam(nxt) - lam(cur)
        eps = model(x, cur)
        if not x0_pred:                  # noise-prediction forms (:578-579, :594-597, :617-621)
            if o == 1:
                x = x - sig(nxt) * h.expm1() * eps
            elif o == 2:
                r1 = 1 / 2
                s1 = inv(cur + r1 * h)
                u1 = x - sig(s1) * (r1 * h).expm1() * eps
                eps_r1 = model(u1, s1)
                x = x - sig(nxt) * h.expm1() * eps - sig(nxt) / (2 * r1) * h.expm1() * (eps_r1 - eps)
            else:
                r1, r2 = 1 / 3, 2 / 3
                s1, s2 = inv(cur + r1 * h), inv(cur + r2 * h)
                u1 = x - sig(s1) * (r1 * h).expm1() * eps
                eps_r1 = model(u1, s1)
                u2 = x - sig(s2) * (r2 * h).expm1() * eps - sig(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1) * (eps_r1 - eps)
                eps_r2 = model(u2, s2)
                x = x - sig(nxt) * h.expm1() * eps - sig(nxt) / r2 * (h.expm1() / h - 1) * (eps_r2 - eps)
            continue
        if o == 1:
            x = sig(nxt) / sig(cur) * x - torch.expm1(-h) * eps
        elif o == 2:
            r1 = 1 / 2
            s1 = inv(cur + r1 * h)
            u1 = sig(s1) / sig(cur) * x - torch.expm1(-r1 * h) * eps
            eps_r1 = fn(u1, sigma=sig(s1))
            x = sig(nxt) / sig(cur) * x - torch.expm1(-h) * eps - 1 / (2 * r1) * torch.expm1(-h) * (eps_r1 - eps)
        else:
            r1, r2 = 1 / 3, 2 / 3
            s1 = inv(cur + r1 * h)
            s2 = inv(cur + r2 * h)
            u1 = sig(s1) / sig(cur) * x - (-r1 * h).expm1() * eps
            eps_r1 = fn(u1, sigma=sig(s1))
            u2 = sig(s2) / sig(cur) * x - (-r2 * h).expm1() * eps + (r2 / r1) * ((-r2 * h).expm1() / (r2 * h) + 1) * (eps_r1 - eps)
            eps_r2 = fn(u2, sigma=sig(s2))
            x = sig(nxt) / sig(cur) * x - torch.expm1(-h) * eps + 1 / r2 * (torch.expm1(-h) / h + 1) * (eps_r2 - eps)
    return x.clamp(-1.0, 1.0)


def dpm2m_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int, reflow: bool = False) -> torch.Tensor:
    """DPM2MSampler ('DPM-Solver++(2M) Karras', sampler_edm.py:1056-1131): num_steps updates, each reading sigmas[i + 1] -- the
    schedule must hold num_steps + 1 entries (with the module's own N-entry schedule the reference raises IndexError on its last
    step; kept).  A final sigma of 0 returns the last denoised estimate (:1098); final clamp.  The class of the same name in
    stochastic_sampler_edm.py:180-259 is the same recurrence plus ``reflow`` (:214-215: the network output read as a velocity,
    denoised = x - output * sigma)."""
    x = sigmas[0] * noise
    old = None
    for i in range(num_steps):
        s_last, s, s_next = sigmas[i - 1], sigmas[i], sigmas[i + 1]
        den = fn(x, sigma=s)
        if reflow:
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
            den_d = (1 + 1 / (2 * r)) * den - (1 / (2 * r)) * old
            x = (t_min / t_max) * x - (-h_d).expm1() * den_d
        old = den
    return x.clamp(-1.0, 1.0)


def lms_coeff(order: int, t, i: int, j: int) -> float:
    """LMSSampler.linear_multistep_coeff (sampler_edm.py:1149-1160): integral over [t_i, t_{i+1}] of the Lagrange basis
    polynomial of node t_{i-j} among t_i .. t_{i-order+1} (scipy quad, epsrel 1e-4, as in the reference)."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def basis(tau):
        prod = 1.0
        for k in range(order):
            if j == k:
                continue
            prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


def lms_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int, order: int = 4) -> torch.Tensor:
    """LMSSampler.forward (sampler_edm.py:1162-1190, 'LMS Karras'): num_steps-1 evaluations, derivative history of up to
    `order` entries, coefficients from the fp32 sigma list on the host, final clamp."""
    t = sigmas.detach().cpu().numpy()
    x = sigmas[0] * noise
    ds: List[torch.Tensor] = []
    for i in range(num_steps - 1):
        d = (x - fn(x, sigma=sigmas[i])) / sigmas[i]
        ds.append(d)
        if len(ds) > order:
            ds.pop(0)
        cur = min(i + 1, order)
        coeffs = [lms_coeff(cur, t, i, j) for j in range(cur)]
        x = x + sum(c * dd for c, dd in zip(coeffs, reversed(ds)))
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


def adpmpp2s_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int, eta: float = 1.0,
                      injected_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ADPMPP2SSampler ('DPM++ 2S a Karras', stochastic_sampler_edm.py:162-178 loop and final clamp, :117-160 step, :29-32
    get_sigmas): num_steps - 1 steps of two evaluations (one when sigma_down is 0: the Euler branch :136-140), fresh noise of
    scale sigma_up after every step whose sigma_next is positive (:158-159).  The constructor's rho is never read.
    ``injected_noise[k]`` replaces the k-th randn_like draw (draws are only consumed by steps with sigma_next > 0)."""
    x = sigmas[0] * noise
    k = 0
    for i in range(num_steps - 1):
        s, s_next = sigmas[i], sigmas[i + 1]
        den = fn(x, sigma=s)
        s_up = min(s_next, eta * (s_next ** 2 * (s ** 2 - s_next ** 2) / s ** 2) ** 0.5)
        s_down = (s_next ** 2 - s_up ** 2) ** 0.5
        if s_down == 0:
            x = x + (x - den) / s * (s_down - s)
        else:
            t, t_next = s.log().neg(), s_down.log().neg()
            h = t_next - t
            sm = t + 0.5 * h
            x_2 = (sm.neg().exp() / t.neg().exp()) * x - (-h * 0.5).expm1() * den
            den_2 = fn(x_2, sigma=sm.neg().exp())
            x = (t_next.neg().exp() / t.neg().exp()) * x - (-h).expm1() * den_2
        if s_next > 0:
            x = x + (injected_noise[k] if injected_noise is not None else torch.randn_like(x)) * s_up
            k += 1
    return x.clamp(-1.0, 1.0)


def unipc_sampler(noise: torch.Tensor, fn: Callable, sigmas: torch.Tensor, num_steps: int, order: int = 2,
                  log_time_spacing: bool = True, x0_pred: bool = True, trace: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """UniPCSampler (sampler_edm.py:807-1053; variant 'bh2'): multistep predictor-corrector, NFE = its step count.  The
    reference only runs on 4-D states (hard-coded einsum 'k,bkchw->bchw', :955); this restatement sums the same terms in
    the same order for any shape.  ``num_steps`` is the constructor argument: the loop makes num_steps steps on a lambda
    grid that is linear between the first and the last sigma (log_time_spacing, :862-866) or num_steps - 1 steps on the sigma
    list itself (:828, :868-870).  x0_pred=False: the model value is (x - D) / sigma (:841-846).  Final clamp (:1053)."""
    steps = num_steps if log_time_spacing else num_steps - 1
    assert steps >= order
    grid, lam, sig, _ = _dpm_grid(sigmas, steps, log_time_spacing)
    model = (lambda x_, g: fn(x_, sigma=sig(g))) if x0_pred else (lambda x_, g: (x_ - fn(x_, sigma=sig(g))) / sig(g))

    def update(x, m_list, g_list, g_cur, ord_, use_corrector):                 # multistep_uni_pc_update, :872-994
        g0, m0 = g_list[-1], m_list[-1]
        h = (lam(g_cur) - lam(g0)).view(-1)
        rks, d1s = [], []
        for i in range(1, ord_):
            rk = (lam(g_list[-(i + 1)]) - lam(g0)) / h
            rks.append(rk)
            d1s.append((m_list[-(i + 1)] - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h if x0_pred else h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = torch.expm1(hh)
        fact = 1
        R, b = [], []
        for i in range(1, ord_ + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / b_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.cat(b)
        rhos_p = None
        if d1s:
            rhos_p = torch.tensor([0.5]) if ord_ == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
        rhos_c = (torch.tensor([0.5]) if ord_ == 1 else torch.linalg.solve(R, b)) if use_corrector else None
        scale = sig(g_cur) if not x0_pred else 1.0
        if x0_pred:
            xt_ = sig(g_cur) / sig(g0) * x - h_phi_1 * m0
        else:
            xt_ = x - sig(g_cur) * h_phi_1 * m0

        def comb(rhos):                                                       # einsum('k,bkchw->bchw'): k ascending
            acc = 0
            for k, d in enumerate(d1s):
                acc = acc + rhos[k] * d
            return acc

        x_t = xt_ - scale * b_h * (comb(rhos_p) if d1s else 0)
        m_t = None
        if use_corrector:
            m_t = model(x_t, g_cur)
            x_t = xt_ - scale * b_h * ((comb(rhos_c[:-1]) if d1s else 0) + rhos_c[-1] * (m_t - m0))
        return x_t, m_t

    x = sigmas[0] * noise
    m_list, g_list = [model(x, grid[0])], [grid[0]]
    for step in range(1, order):                                               # :1013-1022
        x, m = update(x, m_list, g_list, grid[step], step, True)
        g_list.append(grid[step])
        m_list.append(m)
        if trace is not None:
            trace.append(x.clone())
    for step in range(order, steps + 1):                                       # :1025-1051
        x, m = update(x, m_list, g_list, grid[step], min(order, steps + 1 - step), step != steps)
        for i in range(order - 1):
            g_list[i], m_list[i] = g_list[i + 1], m_list[i + 1]
        g_list[-1] = grid[step]
        if step < steps:
            m_list[-1] = m
        if trace is not None:
            trace.append(x.clone())
    return x.clamp(-1.0, 1.0)
