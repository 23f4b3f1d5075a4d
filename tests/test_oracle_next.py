"""CPU: the oracles of the two next hot-path rows (SURVEY.md 8f rows 3 and 4 -- the ADM 2-D U-Net of BASELINE config 4
and the DiffWave ``WaveNetNoise`` of config 5) reproduce fixtures that are outputs of the reference modules themselves
(imported on CPU by oracle/gen_golden_next.py in the build container).  Both have a device path held to these fixtures
(tests/test_adm.py, tests/test_wavenet.py); the constructor variants of the ADM net that are not on the device are pinned here only."""
import os

import numpy as np
import pytest
import torch

from oracle import edm as E, samplers as S, unet2d_oai as A, wavenet as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy
TOL = 2e-6          # fp32, same operations in the same order as the reference: measured 0.0 in the build container


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def sub(t, stride):
    return t.reshape(t.shape[0], -1)[:, ::stride]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "next_golden.npz"))


def adm_variants():
    base = A.config_c4_small()
    kw = base.to_kwargs()
    return {
        "small": base,
        "cls_new": A.ADMConfig(**{**kw, "num_classes": 5, "use_new_attention_order": True, "num_head_channels": 16}),
        "updown": A.ADMConfig(**{**kw, "resblock_updown": True, "use_scale_shift_norm": False}),
        "pool": A.ADMConfig(**{**kw, "conv_resample": False, "attention_resolutions": "32,16", "channel_mult": (1, 1, 2)}),
    }


@pytest.mark.parametrize("tag", ["small", "cls_new", "updown", "pool"])
def test_adm_unet_forward_and_block_outputs(gold, tag):
    """Scale-shift and additive conditioning, legacy and new attention order, conv and pooled / resblock resampling,
    class-conditional with kept and dropped labels."""
    cfg = adm_variants()[tag]
    w = A.generate_weights(cfg, seed=3)
    x, t = T(gold[f"adm_{tag}_x"]), T(gold[f"adm_{tag}_t"])
    classes = T(gold[f"adm_{tag}_classes"]) if cfg.num_classes is not None else None
    taps = {}
    with torch.no_grad():
        y = A.unet2d_forward(w, cfg, x, t, classes=classes, taps=taps)
    assert rel(y, T(gold[f"adm_{tag}_y"])) < TOL
    names = [k[len(f"adm_{tag}_tap_"):] for k in gold.files if k.startswith(f"adm_{tag}_tap_")]
    assert len(names) >= 9
    for k in names:
        assert rel(sub(taps[k], 16), T(gold[f"adm_{tag}_tap_{k}"])) < TOL, k
    if classes is not None:
        with torch.no_grad():
            y0 = A.unet2d_forward(w, cfg, x, t, classes=classes, cond_drop_prob=1.0)
        assert rel(y0, T(gold[f"adm_{tag}_y_null"])) < TOL
        assert rel(y0, y) > 1e-3          # the label matters


def test_adm_structure_of_config4():
    """with the default constructor only the middle block has attention (ds = 16 is never reached)."""
    s = A.structure(A.config_c4())
    kinds = [l.kind for blk in s.input_blocks + s.output_blocks for l in blk]
    assert "attn" not in kinds and [l.kind for l in s.middle] == ["res", "attn", "res"]
    assert len(s.input_blocks) == 12 and len(s.output_blocks) == 12
    specs = A.param_specs(A.config_c4())
    assert len(specs) == 276 and sum(int(np.prod(v[0])) for v in specs.values()) == 70950273
    with pytest.raises(AssertionError):
        A.unet2d_forward({}, A.config_c4(), torch.zeros(1, 1, 8, 8), torch.zeros(1), classes=torch.zeros(1, dtype=torch.long))


@pytest.mark.timeout(300)
def test_adm_config4_full_size(gold):
    """The BASELINE config-4 net itself (71 M parameters) on one 1 x 80 x 256 mel frame block."""
    cfg = A.config_c4()
    w = A.generate_weights(cfg, seed=4)
    with torch.no_grad():
        y = A.unet2d_forward(w, cfg, T(gold["adm_c4_x"]), T(gold["adm_c4_t"]))
    assert y.shape == (1, 1, 80, 256) and rel(y, T(gold["adm_c4_y"])) < TOL


def test_adm_config4_sampler_with_injected_draws(gold):
    """Config 4's sampler: EDMSampler(s_churn=40, s_noise=1.003, s_tmin=0.05, s_tmax=50, num_steps=35) = 69 evaluations,
    on a 4-D state, the reference's randn_like draws injected."""
    cfg = A.config_c4_small()
    w = A.generate_weights(cfg, seed=3)
    calls = {"n": 0}

    def net(xi, ti, **_kw):
        calls["n"] += 1
        return A.unet2d_forward(w, cfg, xi, ti)

    def fn(x, sigma=None, sigmas=None):
        return E.denoise(net, x, 0.5, sigma=sigma, sigmas=sigmas)

    with torch.no_grad():
        y = S.edm_sampler(T(gold["adm_samp_noise"]), fn, T(gold["adm_samp_sigmas"]), 35, s_tmin=0.05, s_tmax=50.0,
                          s_churn=40.0, s_noise=1.003, injected_noise=T(gold["adm_samp_draws"]))
    assert calls["n"] == 69
    assert rel(y, T(gold["adm_samp_y"])) < 5e-5


def test_adm_bf16_storage_mode_and_teacher_forcing():
    cfg = A.config_c4_small()
    w = A.generate_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(2)
    x, t = torch.randn(2, 1, 16, 32, generator=g), torch.tensor([0.3, -0.5])
    t32, t16 = {}, {}
    with torch.no_grad():
        y32 = A.unet2d_forward(w, cfg, x, t, taps=t32)
        y16 = A.unet2d_forward(w, cfg, x, t, taps=t16, storage="bf16")
    assert 1e-4 < A.rel_l2(y16, y32) < 5e-2
    assert set(t16) == set(t32) and "input_blocks.3.1.qkv" in t16 and "middle_block.0.h1" in t16
    forced = {k: v.reshape(v.shape[0], v.shape[1], -1).clone() for k, v in t16.items()}       # the device's [B, C, H*W] tap layout
    errs = {}
    with torch.no_grad():
        A.unet2d_forward(w, cfg, x, t, storage="bf16", force=forced, errs=errs)
    assert max(errs.values()) < 1e-6                      # forcing a run with its own taps changes nothing
    forced["middle_block.0.h1"] = forced["middle_block.0.h1"] + 0.05 * forced["middle_block.0.h1"].flip(-1)
    errs = {}
    with torch.no_grad():
        A.unet2d_forward(w, cfg, x, t, storage="bf16", force=forced, errs=errs)
    bad = {k for k, e in errs.items() if e > 1e-6}
    assert bad == {"middle_block.0.h1", "middle_block.0"}, bad      # the forced tensor and its one consumer


UNIPC_CASES = {"o2_log": dict(order=2, log_time_spacing=True), "o3_log": dict(order=3, log_time_spacing=True),
               "o1_log": dict(order=1, log_time_spacing=True), "o2_sig": dict(order=2, log_time_spacing=False),
               "o3_sig": dict(order=3, log_time_spacing=False), "o2_log_eps": dict(order=2, log_time_spacing=True, x0_pred=False)}


@pytest.mark.parametrize("tag", sorted(UNIPC_CASES))
def test_unipc_sampler_vs_reference_golden(gold, tag):
    """UniPCSampler (sampler_edm.py:807-1053) only runs on 4-D states in the reference: pinned on the small ADM net, 10 steps, orders 1-3,
    both spacings, x0 and noise prediction.  Also: the evaluation count equals the step count, and the plugin's compatibility branch
    (tensor ops around a foreign fn) computes the same."""
    import audiodiffuser_amd as P
    cfg = A.config_c4_small()
    w = A.generate_weights(cfg, seed=3)
    calls = {"n": 0}

    def net(xi, ti, **_kw):
        calls["n"] += 1
        return A.unet2d_forward(w, cfg, xi, ti)

    fn = lambda x, sigma=None, sigmas=None: E.denoise(net, x, 0.5, sigma=sigma, sigmas=sigmas)
    noise, sig = T(gold["adm_samp_noise"]), T(gold["unipc_sigmas"])
    kw = UNIPC_CASES[tag]
    with torch.no_grad():
        y = S.unipc_sampler(noise, fn, sig, 10, **kw)
    assert calls["n"] == (10 if kw["log_time_spacing"] else 9)
    assert rel(y, T(gold[f"unipc_{tag}_y"])) < 2e-5
    compat = P.UniPCSampler(num_steps=10, **kw)
    with torch.no_grad():
        yc = compat(noise, fn=lambda x, net=None, sigma=None, inference=True, cond_scale=1.0, **k: fn(x, sigma=sigma), net=None, sigmas=sig)
    assert rel(yc, y) < 2e-5


def test_timestep_embedding_layout():
    e = A.timestep_embedding(torch.tensor([0.0, 2.0]), 8)
    assert torch.equal(e[0], torch.tensor([1.0, 1, 1, 1, 0, 0, 0, 0]))       # cosines first (:46)
    assert abs(float(e[1, 4]) - np.sin(2.0)) < 1e-6
    assert A.timestep_embedding(torch.tensor([1.0]), 7).shape == (1, 7) and float(A.timestep_embedding(torch.tensor([1.0]), 7)[0, -1]) == 0.0


# ------------------------------------------------------------------------------------------------ WaveNetNoise
@pytest.mark.parametrize("tag", ["small", "c5"])
def test_wavenet_forward_and_layer_taps(gold, tag):
    cfg = {"small": W.config_c5_small, "c5": W.config_c5}[tag]()
    stride = 4 if tag == "small" else 16
    w = W.generate_weights(cfg, seed=5)
    taps = {}
    with torch.no_grad():
        y = W.wavenet_forward(w, cfg, T(gold[f"wn_{tag}_audio"]), T(gold[f"wn_{tag}_step"]), taps=taps)
    assert rel(y, T(gold[f"wn_{tag}_y"])) < TOL
    names = [k[len(f"wn_{tag}_tap_"):] for k in gold.files if k.startswith(f"wn_{tag}_tap_")]
    assert len(names) >= 8
    for k in names:
        assert rel(sub(taps[k], stride), T(gold[f"wn_{tag}_tap_{k}"])) < TOL, k


def test_wavenet_state_dict_layout_and_weight_norm():
    cfg = W.config_c5()
    specs = W.param_specs(cfg)
    assert sum(int(np.prod(v[0])) if v[0] else 1 for v in specs.values()) == 24034379
    keys = list(specs)
    assert keys[:3] == ["input_projection.conv.module.bias", "input_projection.conv.module.weight_g",
                        "input_projection.conv.module.weight_v"]          # bias first: weight is deleted and g, v re-registered (:37-42)
    assert specs["residual_layer.residual_blocks.35.dilated_conv.conv.module.weight_v"][0] == (512, 256, 3)
    assert [cfg.dilation(n) for n in (0, 11, 12, 35)] == [1, 2048, 1, 2048]
    w = W.generate_weights(W.config_c5_small(), seed=5)
    pre = "residual_layer.residual_blocks.2.dilated_conv"
    eff = W.wn_weight(w, pre)
    assert abs(float(torch.norm(eff)) - float(w[f"{pre}.conv.module.weight_g"])) < 1e-4      # ||w|| = g, whole-tensor norm


def test_wavenet_bf16_storage_mode_is_close_and_teacher_forcing_isolates():
    cfg = W.config_c5_small()
    w = W.generate_weights(cfg, seed=5)
    g = torch.Generator().manual_seed(9)
    audio, step = torch.randn(2, 200, generator=g), torch.tensor([0.2, -0.6])
    t32, t16 = {}, {}
    with torch.no_grad():
        y32 = W.wavenet_forward(w, cfg, audio, step, taps=t32)
        y16 = W.wavenet_forward(w, cfg, audio, step, taps=t16, storage="bf16")
    assert 1e-4 < W.rel_l2(y16, y32) < 3e-2
    for k in ("y3", "g5", "skip", "sp"):
        assert W.rel_l2(t16[k], t32[k]) < 3e-2, k
    # forcing a layer's input: that tap reports the mismatch, the next recorded value is computed from the forced one
    forced = {k: v.clone() for k, v in t16.items()}
    forced["y2"] = forced["y2"] + 0.05 * forced["y2"].flip(-1)
    errs = {}
    with torch.no_grad():
        W.wavenet_forward(w, cfg, audio, step, storage="bf16", force=forced, errs=errs)
    bad = {k for k, e in errs.items() if e > 1e-6}
    assert bad == {"y2", "g2", "y3"}, bad


def test_wavenet_adapter_feeds_the_edm_wrapper():
    """No reference caller exists above ``WaveNetNoise.forward`` (it rejects denoise_fn's keyword arguments; the fixture
    report records that), so the adapter is this build's: x[B, 1, T] -> [B, 1, T], extra keyword arguments ignored."""
    cfg = W.config_c5_small()
    w = W.generate_weights(cfg, seed=5)
    net = W.wavenet_net(w, cfg)
    x = torch.randn(2, 1, 128)
    with torch.no_grad():
        d = E.denoise(net, x, 0.5, sigma=1.5)
        ref = W.wavenet_forward(w, cfg, x[:, 0] * (1.5 ** 2 + 0.25) ** -0.5, torch.full((2,), float(np.log(1.5) * 0.25)))
    assert d.shape == x.shape and float(d.abs().max()) <= 1.0
    c_skip, c_out = 0.25 / (1.5 ** 2 + 0.25), 1.5 * 0.5 * (0.25 + 1.5 ** 2) ** -0.5
    assert rel(d, (c_skip * x + c_out * ref).clamp(-1, 1)) < 1e-5


def test_adm_additive_conditioning_alone_vs_reference():
    """use_scale_shift_norm=False without resblock_updown (the form the device serves too): fixture of oracle/gen_golden_adm_add.py."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "adm_add_golden.npz"))
    cfg = A.ADMConfig(**{**A.config_c4_small().to_kwargs(), "use_scale_shift_norm": False})
    w = A.generate_weights(cfg, seed=3)
    assert tuple(w["input_blocks.1.0.emb_layers.1.weight"].shape) == (32, 128)
    taps = {}
    with torch.no_grad():
        y = A.unet2d_forward(w, cfg, T(g["x"]), T(g["t"]), taps=taps)
    assert rel(y, T(g["y"])) < TOL
    names = [k[4:] for k in g.files if k.startswith("tap_")]
    assert len(names) == 9
    for k in names:
        assert rel(sub(taps[k], 16), T(g[f"tap_{k}"])) < TOL, k
