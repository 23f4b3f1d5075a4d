"""Golden vectors for ``use_nearest_upsample=True`` (reference: src/models/backbones/unet1d.py:236-246 -- nn.Upsample(nearest) ->
nn.ReflectionPad1d(1) -> Conv1d(k = 3)): the REFERENCE ``UNet1dBase`` imported on CPU with the generated weights of
``config_tiny_nearest``; checks the oracle restatement against it on every module boundary and writes
``tests/golden/nearest_golden.npz`` + ``nearest_golden_report.json``.

Usage:  python oracle/gen_golden_nearest.py [--check-only]
Test infrastructure only (see oracle/__init__.py)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import import_reference, build_ref_net, rel_err, sub, GOLD   # noqa: E402


def main():
    check_only = "--check-only" in sys.argv
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ref = import_reference()
    from audiodiffuser_amd.config import config_tiny_nearest
    from audiodiffuser_amd.weights import generate_weights, generate_noise
    from oracle import unet1d as O

    cfg = config_tiny_nearest()
    w = generate_weights(cfg, seed=0)
    net = build_ref_net(ref, cfg, w)
    B, L = 2, 256
    x = generate_noise(0, B, L) * 0.7
    t = torch.tensor([-0.9, 0.35], dtype=torch.float32)
    taps_ref = {}
    hooks = []
    for u, up in enumerate(net.unet.upsamples):
        hooks.append(up.upsample.register_forward_hook(lambda m, i, o, u=u: taps_ref.__setitem__(f"up{u}.conv", o.detach().clone())))
    y_ref = net(x, t, cond_drop_prob=0.0)
    for h in hooks:
        h.remove()
    taps_o = {}
    y_o = O.unet1d_forward(w, cfg, x, t, taps=taps_o)
    errs = {k: rel_err(taps_o[k], v) for k, v in taps_ref.items()}
    errs["out"] = rel_err(y_o, y_ref)
    assert len(taps_ref) == cfg.num_layers and max(errs.values()) < 2e-5, errs
    report = {"net_tiny_nearest": errs, "upsample_modules": [type(m).__name__ for m in net.unet.upsamples[0].upsample]}
    print(json.dumps(report))
    if check_only:
        return
    out = {"net_x": x.numpy(), "net_t": t.numpy(), "net_y": y_ref.numpy()}
    for k, v in taps_ref.items():
        out[f"net_tap_{k}"] = sub(v, 7)
    np.savez_compressed(os.path.join(GOLD, "nearest_golden.npz"), **out)
    with open(os.path.join(GOLD, "nearest_golden_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", os.path.join(GOLD, "nearest_golden.npz"), os.path.getsize(os.path.join(GOLD, "nearest_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
