"""Shared helpers of the GPU parity tests: build the HIP net with the deterministic weights, run it,
and compare every recorded activation ("tap") with the CPU oracle on the same inputs."""
from __future__ import annotations

import torch

import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_weights, generate_noise
from oracle import unet1d as O


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))


def make_net(cfg, dtype="fp32", flags=0, seed=0):
    w = generate_weights(cfg, seed=seed)
    net = A.UNet1dBase.from_config(cfg, compute_dtype=dtype, native_flags=flags)
    net.load_state_dict(w, strict=True)
    return net.cuda(), w


def golden_inputs(tag: str):
    B, L = (2, 256) if tag == "tiny" else (2, 2048)
    x = generate_noise(0, B, L) * 0.7
    t = torch.tensor([-0.9, 0.35][:B], dtype=torch.float32)
    return x, t


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    return O.rel_l2(a, b)


def tap_errors_bf16(cfg, x, t, flags=0, chained=True, classes=None):
    """The HIP bf16 (throughput) path against the bf16-STORAGE oracle (oracle/unet1d.py, storage="bf16": the fp32
    restatement with a bf16 rounding wherever the device stores bf16), relative L2 per recorded layer.

    forced[name]  -- teacher-forced: the oracle computes each layer from the DEVICE's own output of the previous layer,
                     so the figure is that one layer's deviation (summation order + the roundings it flips);
    chain[name]   -- the oracle runs free from the same input: deviations compound through the net (only if ``chained``).
    Returns (forced, chain, y_device, y_oracle_forced, y_oracle_chain)."""
    net, w = make_net(cfg, "bf16", flags)
    kw = {} if classes is None else {"classes": classes.cuda()}
    y = net(x.cuda(), t.cuda(), **kw)
    torch.cuda.synchronize()
    hd = net.native(torch.device("cuda", torch.cuda.current_device()))
    dev = {name: hd.tap(name, x.shape[0], y.device).cpu() for name in hd.tap_names()}
    y = y.cpu()
    forced, chain = {}, {}
    with torch.no_grad():
        y_f = O.unet1d_forward(w, cfg, x, t, storage="bf16", force=dev, errs=forced, classes=classes)
        forced["out"] = rel_l2(y, y_f)
        y_c = None
        if chained:
            taps_c = {}
            y_c = O.unet1d_forward(w, cfg, x, t, storage="bf16", taps=taps_c, classes=classes)
            chain = {k: rel_l2(v, taps_c[k]) for k, v in dev.items()}
            chain["out"] = rel_l2(y, y_c)
    missing = set(dev) - set(forced)
    assert not missing, f"device taps the oracle does not record: {missing}"
    return forced, chain, y, y_f, y_c


def tap_errors(cfg, x, t, dtype="fp32", flags=0):
    net, w = make_net(cfg, dtype, flags)
    y = net(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    hd = net.native(torch.device("cuda", torch.cuda.current_device()))
    taps_o = {}
    with torch.no_grad():
        y_o = O.unet1d_forward(w, cfg, x, t, taps=taps_o)
    errs = {}
    for name in hd.tap_names():
        got = hd.tap(name, x.shape[0], y.device).cpu()
        errs[name] = rel_err(got, taps_o[name])
    errs["out"] = rel_err(y.cpu(), y_o)
    return errs, y.cpu(), y_o
