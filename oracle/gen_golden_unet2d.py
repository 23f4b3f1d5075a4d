"""ORACLE tooling (test infrastructure) -- pins oracle/unet2d.py against the reference's ``UNet2dBase`` imported on CPU, and writes
``tests/golden/unet2d_golden.npz`` + ``unet2d_golden_report.json``.

Runs only in the build container (it imports ``/root/reference``); the test-suite reads the fixtures.
Usage:  python oracle/gen_golden_unet2d.py [--check-only]

What it holds the restatement to (all fp32, the same name-keyed weights and inputs on both sides):
  * ``state_dict`` key ORDER and shapes of ``param_specs`` for four constructor variants (the shipped sc09 structure at fixture width,
    the non-memory-efficient / nearest-upsample / plain-init-conv / unconditional variant, the variant without global-context gates,
    final resnet block and middle attention, and the shipped ``diffunet_complex_sc09`` hyper-parameters at full width);
  * the forward output and the output of ``init_conv``, ``init_resnet_block``, every ``downs.i``, ``mid_block``, every ``ups.i`` and
    ``final_res_block`` (forward hooks) for each of them, class-conditional ones also with ``cond_drop_prob=1`` (the null embedding).
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
from oracle.gen_golden import import_reference, rel_err, GOLD   # noqa: E402
from oracle.gen_golden_next import load_into, sub               # noqa: E402


def variants():
    from oracle.unet2d import fixture_variants
    return fixture_variants()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import_reference()
    from src.models.backbones.unet2d import UNet2dBase
    from oracle import unet2d as U

    out, report = {}, {}
    for tag, (cfg, (b, hh, ww)) in variants().items():
        w = U.generate_weights(cfg, seed=5)
        net = load_into(UNet2dBase(**cfg.to_kwargs()), w)          # asserts key order and shapes of param_specs against the module
        g = torch.Generator().manual_seed(300 + len(tag))
        x = torch.randn(b, cfg.channels, hh, ww, generator=g)
        t = torch.tensor([-0.9, 0.4])[:b]
        classes = (torch.arange(b) * 3 + 1) % cfg.num_classes if cfg.num_classes else None
        taps_ref, hooks = {}, []
        first = lambda o: o[0] if isinstance(o, tuple) else o

        def hook(name, m):
            hooks.append(m.register_forward_hook(lambda _m, _i, o, k=name: taps_ref.__setitem__(k, first(o).detach())))
        hook("init_conv", net.init_conv)
        if net.init_resnet_block is not None:
            hook("init_resnet_block", net.init_resnet_block)
        for i, m in enumerate(net.downs):
            hook(f"downs.{i}", m)
        hook("mid_block", net.mid_block)
        for i, m in enumerate(net.ups):
            hook(f"ups.{i}", m)
        if net.final_res_block is not None:
            hook("final_res_block", net.final_res_block)
        with torch.no_grad():
            y_ref = net(x, t, classes=classes)
        for h in hooks:
            h.remove()
        taps = {}
        y = U.unet2d_forward(w, cfg, x, t, classes=classes, taps=taps)
        assert set(taps_ref) == set(taps), (sorted(set(taps_ref) ^ set(taps)))
        errs = {k: rel_err(taps[k], v) for k, v in taps_ref.items()}
        errs["out"] = rel_err(y, y_ref)
        if classes is not None:
            with torch.no_grad():
                y_null = net(x, t, classes=classes, cond_drop_prob=1.0)
            errs["out_null"] = rel_err(U.unet2d_forward(w, cfg, x, t, classes=classes, cond_drop_prob=1.0), y_null)
            assert rel_err(y_null, y_ref) > 1e-3, "the label does not reach the output"
            out[f"u2d_{tag}_y_null"] = y_null.numpy()
            out[f"u2d_{tag}_classes"] = classes.numpy()
        assert max(errs.values()) < 5e-6, (tag, errs)          # fp32 summation-order noise (einsum vs matmul, 47 M parameters deep)
        assert float(y_ref.abs().max()) > 1e-2, "vacuous output"
        report[tag] = {"max_rel_err": max(errs.values()), "taps": len(taps_ref), "tensors": len(w),
                       "params": int(sum(v.numel() for v in w.values())), "input": [b, cfg.channels, hh, ww]}
        out[f"u2d_{tag}_x"] = x.numpy()
        out[f"u2d_{tag}_t"] = t.numpy()
        out[f"u2d_{tag}_y"] = y_ref.numpy()
        stride = 16 if tag != "sc09" else 64
        for k, v in taps_ref.items():
            out[f"u2d_{tag}_tap_{k}"] = sub(v, stride)
        print(tag, report[tag])
        del net
    if args.check_only:
        print(json.dumps(report, indent=1))
        return
    np.savez_compressed(os.path.join(GOLD, "unet2d_golden.npz"), **out)
    with open(os.path.join(GOLD, "unet2d_golden_report.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(GOLD, "unet2d_golden.npz"), sum(v.nbytes for v in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
