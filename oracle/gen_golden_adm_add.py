"""Golden vectors for the ADM UNetModel with ``use_scale_shift_norm=False`` ALONE (additive conditioning, reference:
src/models/backbones/unet2d_oai.py:268-270 -- ``h = out_norm(h + emb_out)``; oracle/gen_golden_next.py holds it only together with
``resblock_updown``): the REFERENCE module imported on CPU with the generated weights of ``config_c4_small``; checks the oracle
against it on every block output and writes ``tests/golden/adm_add_golden.npz`` + ``adm_add_golden_report.json``.

Usage:  python oracle/gen_golden_adm_add.py [--check-only]
Test infrastructure only (see oracle/__init__.py)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import rel_err, GOLD   # noqa: E402
from oracle.gen_golden_next import import_next, load_into, sub   # noqa: E402


def main():
    check_only = "--check-only" in sys.argv
    torch.manual_seed(0)
    torch.set_num_threads(8)
    torch.set_grad_enabled(False)
    ref = import_next()
    from oracle import unet2d_oai as A
    base = A.config_c4_small()
    cfg = A.ADMConfig(**{**base.to_kwargs(), "use_scale_shift_norm": False})
    w = A.generate_weights(cfg, seed=3)
    net = load_into(ref["UNetModel"](**cfg.to_kwargs()), w)
    g = torch.Generator().manual_seed(103)
    x = torch.randn(2, cfg.in_channels, 16, 32, generator=g)
    t = torch.tensor([-0.9, 0.4])
    taps_ref, hooks = {}, []
    for i, m in enumerate(net.input_blocks):
        hooks.append(m.register_forward_hook(lambda _m, _i, o, k=f"input_blocks.{i}": taps_ref.__setitem__(k, o.detach())))
    hooks.append(net.middle_block.register_forward_hook(lambda _m, _i, o: taps_ref.__setitem__("middle_block", o.detach())))
    for i, m in enumerate(net.output_blocks):
        hooks.append(m.register_forward_hook(lambda _m, _i, o, k=f"output_blocks.{i}": taps_ref.__setitem__(k, o.detach())))
    y_ref = net(x, t)
    for h in hooks:
        h.remove()
    taps = {}
    y = A.unet2d_forward(w, cfg, x, t, taps=taps)
    errs = {k: rel_err(taps[k], v) for k, v in taps_ref.items()}
    errs["out"] = rel_err(y, y_ref)
    assert max(errs.values()) < 2e-6 and float(y_ref.abs().max()) > 1e-2, errs
    emb_shape = tuple(net.input_blocks[1][0].emb_layers[1].weight.shape)
    report = {"adm_add": {"max_rel_err": max(errs.values()), "taps": len(taps_ref), "emb_layers.1.weight": list(emb_shape)}}
    print(json.dumps(report))
    if check_only:
        return
    out = {"x": x.numpy(), "t": t.numpy(), "y": y_ref.numpy()}
    for k, v in taps_ref.items():
        out[f"tap_{k}"] = sub(v, 16)
    np.savez_compressed(os.path.join(GOLD, "adm_add_golden.npz"), **out)
    with open(os.path.join(GOLD, "adm_add_golden_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", os.path.join(GOLD, "adm_add_golden.npz"), os.path.getsize(os.path.join(GOLD, "adm_add_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
