"""CPU: oracle/unet2d.py -- the restatement of the reference's Imagen-style ``UNet2dBase`` (src/models/backbones/unet2d.py:622, the network of every
shipped experiment config) -- reproduces fixtures that are outputs of the reference module itself (imported on CPU by oracle/gen_golden_unet2d.py in
the build container).  An oracle-first start: there is no device path behind this network yet, so these tests are the whole of its coverage."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import unet2d as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy
TOL = 5e-6          # fp32; the generator measured <= 2.1e-6 against the reference (einsum vs matmul, rearrange vs pixel_unshuffle)


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def sub(t, stride):
    return t.reshape(t.shape[0], -1)[:, ::stride]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "unet2d_golden.npz"))


@pytest.mark.parametrize("tag", ["small", "nomem", "nogca", "sc09"])
def test_unet2d_forward_and_block_outputs(gold, tag):
    """Memory-efficient (pre-downsample, initial resnet block, pixel-shuffle upsampling) and plain (post-downsample, the Parallel 3x3 + 1x1 conv of the
    last level, nearest upsampling) layouts, cross-embed and plain initial conv, global-context gates on and off, one and two transformer layers,
    class-conditional with kept and dropped labels; ``sc09`` = the shipped diffunet_complex_sc09 hyper-parameters at full width (47.3 M parameters)."""
    cfg, (b, hh, ww) = U.fixture_variants()[tag]
    w = U.generate_weights(cfg, seed=5)
    x, t = T(gold[f"u2d_{tag}_x"]), T(gold[f"u2d_{tag}_t"])
    assert tuple(x.shape) == (b, cfg.channels, hh, ww)
    classes = T(gold[f"u2d_{tag}_classes"]) if cfg.num_classes else None
    taps = {}
    with torch.no_grad():
        y = U.unet2d_forward(w, cfg, x, t, classes=classes, taps=taps)
    assert rel(y, T(gold[f"u2d_{tag}_y"])) < TOL
    names = [k[len(f"u2d_{tag}_tap_"):] for k in gold.files if k.startswith(f"u2d_{tag}_tap_")]
    assert set(names) == set(taps) and len(names) >= 7
    stride = 16 if tag != "sc09" else 64
    for k in names:
        assert rel(sub(taps[k], stride), T(gold[f"u2d_{tag}_tap_{k}"])) < TOL, k
    if classes is not None:
        with torch.no_grad():
            y0 = U.unet2d_forward(w, cfg, x, t, classes=classes, cond_drop_prob=1.0)
        assert rel(y0, T(gold[f"u2d_{tag}_y_null"])) < TOL
        assert rel(y0, y) > 1e-3                           # the label reaches the output


def test_state_dict_layout_matches_the_generator_report():
    """The generator loads ``generate_weights`` into the reference module with ``strict=True`` after asserting key ORDER and shapes; its report holds the
    tensor and parameter counts it saw: the layout of ``param_specs`` has not drifted since."""
    rep = json.load(open(os.path.join(ROOT, "tests", "golden", "unet2d_golden_report.json")))
    for tag, (cfg, _) in U.fixture_variants().items():
        specs = U.param_specs(cfg)
        n = sum(int(np.prod(shape)) for shape, _ in specs.values())
        assert (len(specs), n) == (rep[tag]["tensors"], rep[tag]["params"]), tag
        assert rep[tag]["max_rel_err"] < TOL
    assert rep["sc09"]["params"] == 47279260


def test_unsupported_paths_raise_rather_than_guess():
    cfg = U.config_sc09_small()
    w = {}
    with pytest.raises(AssertionError):
        U.UNet2dConfig(dim=96).check()                                     # the constructor's ``dim > 100`` (:671)
    with pytest.raises(AssertionError):
        U.UNet2dConfig(dim=128, num_classes=3, cond_dim=64).check()        # t + classes_emb needs equal widths
    with pytest.raises(AssertionError):
        U.unet2d_forward(U.generate_weights(cfg, 0), cfg, torch.zeros(1, 2, 16, 8), torch.zeros(1))      # class-conditional net without labels
    assert not hasattr(U.UNet2dConfig(), "cond_on_text") and not hasattr(U.UNet2dConfig(), "use_condition_block") and w == {}
