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
