"""ORACLE tooling (test infrastructure) -- pins oracle/unet2d_oai.py and oracle/wavenet.py against the reference
modules imported on CPU, and writes ``tests/golden/next_golden.npz`` + ``next_golden_report.json``.

Runs only in the build container (it imports ``/root/reference``); the GPU box and the test-suite read the fixtures.
Usage:  python oracle/gen_golden_next.py [--check-only]

What it holds the restatements to (all fp32, same weights / inputs on both sides):
  * ADM 2-D U-Net: state_dict key order + shapes for four constructor variants; forward output and the output of every
    input / middle / output block (forward hooks) for the fixture nets, incl. class-conditional + new attention order +
    resblock up/down + non-scale-shift variants; the BASELINE config-4 net (default constructor, 1 x 80 x 256) once;
    a 35-step churn EDM sampler run (config 4's sampler arguments) with injected draws on the small net.
  * WaveNetNoise: state_dict key order + shapes; forward output and per-layer taps for the fixture net and for the
    default 36-layer / 256-channel net.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import import_reference, rel_err, REF, GOLD   # noqa: E402


def import_next():
    import_reference()
    from src.models.backbones.unet2d_oai import UNetModel
    from src.models.backbones.wavenet import WaveNetNoise
    from src.models.components.diffusion import EluDiffusion
    from src.models.components.sampler_edm import EDMSampler, UniPCSampler
    from src.models.components.scheduler import KarrasSchedule
    return dict(UNetModel=UNetModel, WaveNetNoise=WaveNetNoise, EluDiffusion=EluDiffusion, EDMSampler=EDMSampler,
                KarrasSchedule=KarrasSchedule, UniPCSampler=UniPCSampler)


def load_into(net, weights):
    sd = net.state_dict()
    assert list(sd.keys()) == list(weights.keys()), ("state_dict key order/name mismatch",
                                                      [a for a, b in zip(sd.keys(), weights.keys()) if a != b][:5])
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(weights[k].shape), (k, tuple(v.shape), tuple(weights[k].shape))
    net.load_state_dict(weights, strict=True)
    return net.eval()


def sub(t, stride):
    return t.reshape(t.shape[0], -1)[:, ::stride].contiguous().numpy()


def adm_variants():
    from oracle.unet2d_oai import ADMConfig, config_c4_small
    base = config_c4_small()
    return {
        "small": base,
        "cls_new": ADMConfig(**{**base.to_kwargs(), "num_classes": 5, "use_new_attention_order": True, "num_head_channels": 16}),
        "updown": ADMConfig(**{**base.to_kwargs(), "resblock_updown": True, "use_scale_shift_norm": False}),
        "pool": ADMConfig(**{**base.to_kwargs(), "conv_resample": False, "attention_resolutions": "32,16",
                             "channel_mult": (1, 1, 2)}),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_next()
    from oracle import unet2d_oai as A, wavenet as W, edm as E, samplers as S

    out, report = {}, {}

    # ================================================================= ADM 2-D U-Net
    report["adm"] = {}
    for tag, cfg in adm_variants().items():
        w = A.generate_weights(cfg, seed=3)
        net = load_into(ref["UNetModel"](**cfg.to_kwargs()), w)
        g = torch.Generator().manual_seed(100 + len(tag))
        b, hh, ww = 2, 16, 32
        x = torch.randn(b, cfg.in_channels, hh, ww, generator=g)
        t = torch.tensor([-0.9, 0.4])[:b]
        classes = torch.tensor([1, 4]) if cfg.num_classes is not None else None
        taps_ref = {}
        hooks = []
        for i, m in enumerate(net.input_blocks):
            hooks.append(m.register_forward_hook(lambda _m, _i, o, k=f"input_blocks.{i}": taps_ref.__setitem__(k, o.detach())))
        hooks.append(net.middle_block.register_forward_hook(lambda _m, _i, o: taps_ref.__setitem__("middle_block", o.detach())))
        for i, m in enumerate(net.output_blocks):
            hooks.append(m.register_forward_hook(lambda _m, _i, o, k=f"output_blocks.{i}": taps_ref.__setitem__(k, o.detach())))
        with torch.no_grad():
            y_ref = net(x, t, classes=classes) if classes is not None else net(x, t)
        for h in hooks:
            h.remove()
        with torch.no_grad():
            y_null = net(x, t, classes=classes, cond_drop_prob=1.0) if classes is not None else None
        taps = {}
        y = A.unet2d_forward(w, cfg, x, t, classes=classes, taps=taps)
        errs = {k: rel_err(taps[k], v) for k, v in taps_ref.items()}
        errs["out"] = rel_err(y, y_ref)
        if y_null is not None:
            errs["out_null"] = rel_err(A.unet2d_forward(w, cfg, x, t, classes=classes, cond_drop_prob=1.0), y_null)
            out[f"adm_{tag}_y_null"] = y_null.numpy()
        assert set(taps_ref) <= set(taps), sorted(set(taps_ref) - set(taps))
        assert max(errs.values()) < 2e-6, (tag, errs)
        assert float(y_ref.abs().max()) > 1e-2, "vacuous output"
        report["adm"][tag] = {"max_rel_err": max(errs.values()), "taps": len(taps_ref),
                              "params": int(sum(v.numel() for v in w.values()))}
        out[f"adm_{tag}_x"] = x.numpy()
        out[f"adm_{tag}_t"] = t.numpy()
        out[f"adm_{tag}_y"] = y_ref.numpy()
        if classes is not None:
            out[f"adm_{tag}_classes"] = classes.numpy()
        for k, v in taps_ref.items():
            out[f"adm_{tag}_tap_{k}"] = sub(v, 16)
        del net

    # ---- BASELINE config 4: default constructor at 1 x 80 x 256 --------------------------------------------
    cfg4 = A.config_c4()
    w4 = A.generate_weights(cfg4, seed=4)
    net4 = load_into(ref["UNetModel"](**cfg4.to_kwargs()), w4)
    g = torch.Generator().manual_seed(44)
    x4 = torch.randn(1, 1, 80, 256, generator=g)
    t4 = torch.tensor([0.3])
    with torch.no_grad():
        y4_ref = net4(x4, t4)
    y4 = A.unet2d_forward(w4, cfg4, x4, t4)
    e4 = rel_err(y4, y4_ref)
    assert e4 < 2e-6, e4
    report["adm"]["c4"] = {"max_rel_err": e4, "params": int(sum(v.numel() for v in w4.values())),
                           "tensors": len(w4)}
    out["adm_c4_x"] = x4.numpy()
    out["adm_c4_t"] = t4.numpy()
    out["adm_c4_y"] = y4_ref.numpy()
    del net4

    # ---- config 4's sampler (35-step churn EDM, 69 evaluations) on the small net, injected draws -------------
    cfg = adm_variants()["small"]
    w = A.generate_weights(cfg, seed=3)
    net = load_into(ref["UNetModel"](**cfg.to_kwargs()), w)
    diff = ref["EluDiffusion"](sigma_data=0.5)
    n_steps = 35
    sampler = ref["EDMSampler"](s_tmin=0.05, s_tmax=50.0, s_churn=40.0, s_noise=1.003, num_steps=n_steps)
    sigmas = ref["KarrasSchedule"](sigma_min=0.002, sigma_max=80.0, rho=7.0, num_steps=n_steps)()
    g = torch.Generator().manual_seed(45)
    noise = torch.randn(2, 1, 16, 32, generator=g)
    draws = torch.randn(n_steps, 2, 1, 16, 32, generator=g)
    calls = {"n": 0}

    def fn_ref(x, net=None, sigma=None, sigmas=None, **kw):
        calls["n"] += 1
        return diff.denoise_fn(x, net=net, sigma=sigma, sigmas=sigmas, **kw)

    it = iter(draws)
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: next(it)
    try:
        with torch.no_grad():
            xs_ref = sampler(noise, fn=fn_ref, net=net, sigmas=sigmas)
    finally:
        torch.randn_like = orig

    def net_o(xi, ti, **_kw):
        return A.unet2d_forward(w, cfg, xi, ti)

    def fn_o(x, sigma=None, sigmas=None):
        return E.denoise(net_o, x, 0.5, sigma=sigma, sigmas=sigmas)

    xs = S.edm_sampler(noise, fn_o, sigmas, n_steps, s_tmin=0.05, s_tmax=50.0, s_churn=40.0, s_noise=1.003,
                       injected_noise=draws)
    es = rel_err(xs, xs_ref)
    assert calls["n"] == 2 * n_steps - 1 and es < 5e-5, (calls, es)
    report["adm"]["sampler_c4_small"] = {"nfe": calls["n"], "max_rel_err": es}
    out["adm_samp_noise"] = noise.numpy()
    out["adm_samp_draws"] = draws.numpy()
    out["adm_samp_sigmas"] = sigmas.numpy()
    out["adm_samp_y"] = xs_ref.numpy()

    # ---- UniPCSampler (sampler_edm.py:807-1053): only runs on 4-D states in the reference, so it is pinned here, on the small ADM net -----
    report["unipc"] = {}
    sig10 = ref["KarrasSchedule"](sigma_min=0.002, sigma_max=80.0, rho=7.0, num_steps=10)()
    out["unipc_sigmas"] = sig10.numpy()

    def fn_plain(x, net=None, sigma=None, sigmas=None, **kw):
        return diff.denoise_fn(x, net=net, sigma=sigma, sigmas=sigmas, **kw)

    for tag, kw in (("o2_log", dict(order=2, log_time_spacing=True)), ("o3_log", dict(order=3, log_time_spacing=True)),
                    ("o1_log", dict(order=1, log_time_spacing=True)), ("o2_sig", dict(order=2, log_time_spacing=False)),
                    ("o3_sig", dict(order=3, log_time_spacing=False)), ("o2_log_eps", dict(order=2, log_time_spacing=True, x0_pred=False))):
        smp = ref["UniPCSampler"](num_steps=10, **kw)
        with torch.no_grad():
            yr = smp(noise, fn=fn_plain, net=net, sigmas=sig10)
            yo = S.unipc_sampler(noise, fn_o, sig10, 10, **kw)
        e = rel_err(yo, yr)
        assert torch.isfinite(yr).all() and e < 2e-5, (tag, e)
        report["unipc"][tag] = e
        out[f"unipc_{tag}_y"] = yr.numpy()
    del net

    # ================================================================= WaveNetNoise
    report["wavenet"] = {}
    for tag, cfg, b, tlen, stride in (("small", W.config_c5_small(), 2, 300, 4), ("c5", W.config_c5(), 1, 1024, 16)):
        w = W.generate_weights(cfg, seed=5)
        net = load_into(ref["WaveNetNoise"](**cfg.to_kwargs()), w)
        g = torch.Generator().manual_seed(55 + b)
        audio = torch.randn(b, tlen, generator=g)
        step = torch.tensor([-1.1, 0.7])[:b]
        taps_ref = {}
        hooks = []
        for n, blk in enumerate(net.residual_layer.residual_blocks):
            hooks.append(blk.dilated_conv.register_forward_hook(
                lambda _m, i, _o, k=f"y{n}": taps_ref.__setitem__(k, i[0].detach())))
            hooks.append(blk.output_projection.register_forward_hook(
                lambda _m, i, _o, k=f"g{n}": taps_ref.__setitem__(k, i[0].detach())))
        hooks.append(net.residual_layer.register_forward_hook(lambda _m, _i, o: taps_ref.__setitem__("skip", o.detach())))
        hooks.append(net.output_projection.register_forward_hook(lambda _m, i, _o: taps_ref.__setitem__("sp", i[0].detach())))
        with torch.no_grad():
            y_ref = net(audio, step)
        for h in hooks:
            h.remove()
        taps = {}
        y = W.wavenet_forward(w, cfg, audio, step, taps=taps)
        errs = {k: rel_err(taps[k], v) for k, v in taps_ref.items()}
        errs["out"] = rel_err(y, y_ref)
        assert len(taps_ref) == 2 * cfg.residual_layers + 2
        assert max(errs.values()) < 5e-6, (tag, max(errs.values()), max(errs, key=errs.get))
        assert float(y_ref.abs().max()) > 1e-2, "vacuous output"
        # the reference rejects the keyword arguments denoise_fn passes: no caller above forward() exists
        try:
            net(audio.unsqueeze(1), step, cond_drop_prob=0.0)
            rejected = False
        except TypeError:
            rejected = True
        assert rejected
        report["wavenet"][tag] = {"max_rel_err": max(errs.values()), "taps": len(taps_ref),
                                  "params": int(sum(v.numel() for v in w.values())),
                                  "forward_rejects_denoise_fn_kwargs": rejected}
        out[f"wn_{tag}_audio"] = audio.numpy()
        out[f"wn_{tag}_step"] = step.numpy()
        out[f"wn_{tag}_y"] = y_ref.numpy()
        keep = list(taps_ref) if tag == "small" else ["y0", "g0", "y12", "g17", "y35", "g35", "skip", "sp"]
        for k in keep:
            out[f"wn_{tag}_tap_{k}"] = sub(taps_ref[k], stride)
        del net

    print(json.dumps(report, indent=1))
    if args.check_only:
        return
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "next_golden.npz"), **out)
    with open(os.path.join(GOLD, "next_golden_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", os.path.join(GOLD, "next_golden.npz"), os.path.getsize(os.path.join(GOLD, "next_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
