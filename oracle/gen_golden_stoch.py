"""Golden vectors for the rest of the reference's stochastic_sampler_edm.py: TEST INFRASTRUCTURE.

Runs the imported reference (this container only; /root/reference is absent on the GPU box) and checks the oracle's
restatements against it before writing tests/golden/stoch_golden.npz:

  * ADPMPP2SSampler ('DPM++ 2S a Karras', stochastic_sampler_edm.py:102-178) with recorded randn_like draws, eta 1.0 and 0.6,
    on a Karras schedule and on one that ends in 0 (the Euler branch :136-140 and the skipped last draw :158);
  * the dynamic threshold of EluDiffusion (components/utils.py:19-33): denoise_fn at three noise levels and an 8-step Heun run, q = 0.95 / 0.5;
  * DPM2MSampler of the same file (:180-259) with reflow=True (its reflow=False path is sampler_edm.py's DPM2MSampler, pinned
    by oracle/gen_golden.py section 9).

DPMPPSDESampler (:261-345) is not pinned: it needs torchsde's BrownianTree (absent here) and its forward() has no return
statement -- there is no output to compare.

    python oracle/gen_golden_stoch.py [--check-only]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import import_reference, build_ref_net, rel_err, GOLD   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    from src.models.components.stochastic_sampler_edm import ADPMPP2SSampler, DPM2MSampler
    from audiodiffuser_amd.config import config_tiny
    from audiodiffuser_amd.weights import generate_weights, generate_noise
    from oracle import edm as E, samplers as S

    cfg = config_tiny()
    w = generate_weights(cfg, seed=0)
    net = build_ref_net(ref, cfg, w)
    diff = ref["EluDiffusion"](sigma_data=0.2)
    fn_o = E.make_denoiser(w, cfg, 0.2)
    noise = generate_noise(70, 2, 256)
    sg = E.karras_sigmas(0.002, 80.0, 7.0, 10)
    sg0 = torch.cat([E.karras_sigmas(0.002, 80.0, 7.0, 9), torch.zeros(1)])
    sg11 = E.karras_sigmas(0.002, 80.0, 7.0, 11)
    out, report = {}, {}

    def run_recorded(sampler, seed0, sigmas):
        draws = []
        real = torch.randn_like

        def rec(x, *a, **k):
            g = torch.Generator(); g.manual_seed(seed0 + len(draws))
            z = torch.randn(x.shape, generator=g, dtype=x.dtype); draws.append(z); return z
        torch.randn_like = rec
        try:
            with torch.no_grad():
                y = sampler(noise, fn=diff.denoise_fn, net=net, sigmas=sigmas)
        finally:
            torch.randn_like = real
        return y, (torch.stack(draws) if draws else None)

    for tag, eta, sigmas, ndraws in (("e1", 1.0, sg, 9), ("e06", 0.6, sg, 9), ("e1_zero", 1.0, sg0, 8)):
        y, inj = run_recorded(ADPMPP2SSampler(num_steps=10, eta=eta), 9400, sigmas)
        assert inj.shape[0] == ndraws, (tag, inj.shape)
        with torch.no_grad():
            yo = S.adpmpp2s_sampler(noise, fn_o, sigmas, 10, eta=eta, injected_noise=inj)
        report[f"adpmpp2s_{tag}"] = rel_err(yo, y)
        out[f"smp_adpmpp2s_{tag}_final"] = y.numpy()
    for tag, sigmas in (("k11", sg11), ("k10_zero", torch.cat([sg, torch.zeros(1)]))):
        with torch.no_grad():
            y = DPM2MSampler(num_steps=10, reflow=True)(noise, fn=diff.denoise_fn, net=net, sigmas=sigmas)
            yo = S.dpm2m_sampler(noise, fn_o, sigmas, 10, reflow=True)
            # reflow=False of this class is sampler_edm.py's DPM2MSampler
            ya = DPM2MSampler(num_steps=10)(noise, fn=diff.denoise_fn, net=net, sigmas=sigmas)
            yb = ref["DPM2MSampler"](num_steps=10)(noise, fn=diff.denoise_fn, net=net, sigmas=sigmas)
        assert torch.equal(ya, yb)
        report[f"dpm2m_reflow_{tag}"] = rel_err(yo, y)
        out[f"smp_dpm2m_reflow_{tag}_final"] = y.numpy()
    # ---- dynamic thresholding (components/utils.py:19-33 via diffusion.py:61): denoise_fn at three noise levels and an 8-step Heun run ----
    for q in (0.95, 0.5):
        diff_q = ref["EluDiffusion"](sigma_data=0.2, dynamic_threshold=q)
        fn_q = E.make_denoiser(w, cfg, 0.2, dynamic_threshold=q)
        x = 3.0 * generate_noise(71, 2, 256)
        for sv in (2.5, 0.4, 0.02):
            with torch.no_grad():
                y = diff_q.denoise_fn(x, net=net, sigma=sv, inference=True)
                yo = fn_q(x, sigma=sv)
            report[f"dyn_q{q}_s{sv}"] = rel_err(yo, y)
            out[f"dyn_q{q}_s{sv}"] = y.numpy()
        sg8 = E.karras_sigmas(0.002, 80.0, 7.0, 8)
        with torch.no_grad():
            y = ref["EDMSampler"](s_churn=0.0, num_steps=8)(noise, fn=diff_q.denoise_fn, net=net, sigmas=sg8)
            yo = S.edm_sampler(noise, fn_q, sg8, 8, s_churn=0.0)
        report[f"dyn_q{q}_heun8"] = rel_err(yo, y)
        out[f"dyn_q{q}_heun8"] = y.numpy()
    out["dyn_x"] = (3.0 * generate_noise(71, 2, 256)).numpy()
    assert all(np.isfinite(v).all() for v in out.values())
    assert max(report.values()) < 5e-4, report
    print(json.dumps(report, indent=1))
    if args.check_only:
        return
    np.savez_compressed(os.path.join(GOLD, "stoch_golden.npz"), **out)
    with open(os.path.join(GOLD, "stoch_golden_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", os.path.join(GOLD, "stoch_golden.npz"))


if __name__ == "__main__":
    main()
