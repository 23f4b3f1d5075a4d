"""ADM-style 2-D U-Net ``UNetModel`` (BASELINE configs[3]): plugin contract on CPU; on the GPU the HIP path
through the C ABI in fp32 mode against the REFERENCE's own outputs (fixtures of oracle/gen_golden_next.py): network output, every
block output, the default 71 M-parameter net at 1 x 80 x 256, and config 4's 35-step churn sampler (69 evaluations, injected draws);
bf16 mode against fp32 as a storage-precision figure."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import audiodiffuser_amd as A
from audiodiffuser_amd import _lib
from audiodiffuser_amd.adm_config import param_specs, generate_weights
from oracle import unet2d_oai as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy
FP32_TOL = 1e-3        # north-star bar, relative to the reference
FP32_TIGHT = 5e-5      # exact-fp32 FMA chains in another summation order (measured ~2e-6)
BF16_CONV_TOL = 5e-4   # (measured: worst 1.8e-4, median 3.7e-5 over the 94 stored tensors of the config-4 net) one launch of the bf16 path from the device's own input against the bf16-storage oracle (relative L2): summation order + the
                       # roundings it flips; a conv with a residual rounds twice (tile, then tile + residual)
BF16_ATT_TOL = 1e-3    # (measured 6e-5) the attention launch (softmax and P V in fp32 on the vector ALUs, bf16 in / out)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "next_golden.npz"))


def make(cfg, dtype="fp32", seed=3):
    w = generate_weights(cfg, seed=seed)
    net = A.UNetModel.from_config(cfg, compute_dtype=dtype)
    net.load_state_dict(w, strict=True)
    return net, w


def test_plugin_state_dict_contract_and_refusals():
    cfg = A.config_c4()
    net = A.UNetModel(in_channels=1, out_channels=1)                        # every other argument the reference's default
    sd, specs = net.state_dict(), param_specs(cfg)
    assert list(sd) == list(specs) and all(tuple(sd[k].shape) == specs[k][0] for k in specs)
    assert sum(p.numel() for p in net.parameters()) == 70950273
    assert float(sd["out.2.weight"].abs().max()) == 0.0 and float(sd["middle_block.1.proj_out.weight"].abs().max()) == 0.0   # zero_module
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 16, 32), torch.zeros(1))
    add = A.UNetModel(use_scale_shift_norm=False, in_channels=1, out_channels=1)        # additive conditioning: Linear(4 mc, cout), not 2 cout
    assert tuple(add.state_dict()["input_blocks.1.0.emb_layers.1.weight"].shape) == (128, 512)
    with pytest.raises(NotImplementedError):
        A.UNetModel(class_embed_dim=8)
    with pytest.raises(ValueError, match="multiple of 64 channels: input_blocks.1.0"):    # the bf16 routes' constraint is reported at construction
        A.UNetModel(model_channels=48, compute_dtype="bf16")
    A.UNetModel(model_channels=48, compute_dtype="fp32")
    assert C.sizeof(_lib.AdfAdmConfig) == 4 * (4 + 1 + 8 + 1 + 8 + 8)
    c = _lib.make_adm_config(cfg, _lib.DTYPE_BF16)
    assert (c.n_mult, list(c.channel_mult)[:4], c.n_attention_ds, c.attention_ds[0]) == (4, [1, 2, 2, 4], 1, 16)


@pytest.mark.gpu
def test_fp32_forward_and_block_outputs_vs_reference_golden(gold):
    cfg = A.config_c4_small()
    net, _ = make(cfg)
    net = net.cuda()
    x, t = T(gold["adm_small_x"]), T(gold["adm_small_t"])
    y = net(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    assert y.shape == x.shape
    e = rel(y.cpu(), T(gold["adm_small_y"]))
    assert e < FP32_TIGHT < FP32_TOL, e
    hd = net.native(torch.device("cuda", torch.cuda.current_device()))
    names = hd.tap_names()
    blocks = [k for k in names if f"adm_small_tap_{k}" in gold.files]
    assert len(blocks) == 9 and len(names) > 30          # the reference's nine block outputs + every tensor the device stores
    for k in blocks:
        tap = hd.tap(k, 2, torch.device("cuda")).cpu()
        et = rel(tap.reshape(2, -1)[:, ::16], T(gold[f"adm_small_tap_{k}"]))
        assert et < FP32_TIGHT, (k, et)


@pytest.mark.gpu
def test_fp32_new_attention_order_vs_oracle():
    """use_new_attention_order=True (no row permutation of the qkv weights) with per-head width 16."""
    cfg = A.ADMConfig(**{**A.config_c4_small().to_kwargs(), "use_new_attention_order": True, "num_head_channels": 16})
    net, w = make(cfg, seed=8)
    g = torch.Generator().manual_seed(2)
    x, t = torch.randn(3, 1, 16, 32, generator=g), torch.tensor([0.2, -0.7, 1.0])
    with torch.no_grad():
        ref = O.unet2d_forward(w, cfg, x, t)
    assert rel(net.cuda()(x.cuda(), t.cuda()).cpu(), ref) < FP32_TIGHT


@pytest.mark.gpu
def test_fp32_larger_batch_and_image_vs_oracle():
    """B = 8 at 32 x 64: enough tiles for the 4 x 32-pixel spatial route of the 3x3 convs at the first level (the fixtures above run the
    2 x 32 route and, where the width is not a multiple of 32, the per-tap gather kernel)."""
    cfg = A.config_c4_small()
    net, w = make(cfg, seed=9)
    g = torch.Generator().manual_seed(5)
    x, t = torch.randn(8, 1, 32, 64, generator=g), torch.linspace(-1.0, 1.0, 8)
    with torch.no_grad():
        ref = O.unet2d_forward(w, cfg, x, t)
    assert rel(net.cuda()(x.cuda(), t.cuda()).cpu(), ref) < FP32_TIGHT


@pytest.mark.gpu
@pytest.mark.parametrize("tile", ["0", "1"])
def test_both_conv_routes_in_a_child_process(tile):
    """ADF_CONV2D_TILE=0 sends every 3x3 conv through the per-tap gather kernel, 1 (default) the same-size ones through the spatial-tile
    kernel; the switch is read once per process."""
    import json, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag", "gpu_adm_report.py")], capture_output=True, text=True,
                       env=dict(os.environ, ADF_CONV2D_TILE=tile), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["route_tile"] == tile and rep["max_rel"] < FP32_TIGHT, rep


@pytest.mark.gpu
def test_ten_row_images_take_the_160_pixel_tiles():
    """H = 10 (the 80-row mel block three levels down) at a batch that fills the chip: the 5 x 32-pixel spatial tiles, fp32 against the oracle and
    bf16 teacher-forced against the bf16-storage oracle; the launcher's route lines prove the kernel."""
    import json, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag", "gpu_adm_t5_report.py")], capture_output=True, text=True,
                       env=dict(os.environ, ADF_C2_TRACE="1"), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sum(1 for l in r.stderr.splitlines() if l.startswith("[adf conv2d] t5 ")) >= 6, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["fp32_max_rel"] < FP32_TIGHT, rep
    assert rep["bf16_taps"] > 15 and not rep["bf16_missing"], rep
    assert rep["bf16_worst_conv"] < BF16_CONV_TOL and rep["bf16_worst_att"] < BF16_ATT_TOL, rep


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_config4_full_size_fp32_and_bf16(gold):
    """The BASELINE config-4 network itself (default constructor, 1 x 80 x 256): fp32 against the reference's output; bf16 against
    fp32 as a storage-precision figure."""
    cfg = A.config_c4()
    net, _ = make(cfg, seed=4)
    x, t = T(gold["adm_c4_x"]), T(gold["adm_c4_t"])
    y = net.cuda()(x.cuda(), t.cuda()).cpu()
    assert rel(y, T(gold["adm_c4_y"])) < FP32_TIGHT
    net16, _ = make(cfg, "bf16", seed=4)
    y16 = net16.cuda()(x.cuda(), t.cuda()).cpu()
    e = float((y16 - y).norm() / y.norm())
    assert e < 5e-2, e


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bf16_every_stored_tensor_vs_bf16_storage_oracle():
    """bf16 (throughput) mode of the BASELINE config-4 network on one 1 x 80 x 256 block (the spatial-tile kernel, the gather kernel for the
    resampling and 1x1 convs, attention at 320 tokens).  Teacher-forced: the oracle computes every stored tensor from the DEVICE's own
    inputs of that launch."""
    cfg = A.config_c4()
    net, w = make(cfg, "bf16", seed=4)
    g = torch.Generator().manual_seed(12)
    x, t = torch.randn(1, 1, 80, 256, generator=g), torch.tensor([0.25])
    net = net.cuda()
    y = net(x.cuda(), t.cuda()).cpu()
    hd = net.native(torch.device("cuda", torch.cuda.current_device()))
    taps = {k: hd.tap(k, 1, torch.device("cuda")).cpu() for k in hd.tap_names()}
    errs = {}
    with torch.no_grad():
        y_f = O.unet2d_forward(w, cfg, x, t, storage="bf16", force=taps, errs=errs)
        y_32 = O.unet2d_forward(w, cfg, x, t)
    assert set(errs) == set(taps) and len(errs) > 80
    worst = max(errs, key=errs.get)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    import json
    with open(os.path.join(ROOT, "gpurun_out", "adm_bf16_parity_report.json"), "w") as f:
        json.dump({"taps": len(errs), "worst": worst, "worst_rel_l2": errs[worst], "median_rel_l2": sorted(errs.values())[len(errs) // 2],
                   "att": {k: v for k, v in errs.items() if k.endswith(".att")}, "out_vs_forced": O.rel_l2(y, y_f),
                   "bf16_vs_fp32_oracle": O.rel_l2(y, y_32)}, f, indent=1)
    for k, e in errs.items():
        assert e < (BF16_ATT_TOL if k.endswith(".att") else BF16_CONV_TOL), (k, e, worst, errs[worst])
    assert O.rel_l2(y, y_f) < BF16_CONV_TOL, O.rel_l2(y, y_f)         # the last conv from the device's last block output
    assert O.rel_l2(y, y_32) < 5e-2, O.rel_l2(y, y_32)               # free-running bf16 vs the fp32 reference arithmetic: storage precision


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_additive_conditioning_vs_reference_golden_and_oracle(dtype):
    """UNetModel(use_scale_shift_norm=False) (unet2d_oai.py:268-270: h = out_norm(h + emb_out)): on the device the embedding is a per-sample addend to
    conv1's bias.  fp32 against the reference's own outputs (fixture of oracle/gen_golden_adm_add.py: network output and the nine block outputs);
    bf16 teacher-forced per stored tensor against the bf16-storage oracle (64-channel widths for the bf16 conv routes)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "adm_add_golden.npz"))
    if dtype == "fp32":
        cfg = O.ADMConfig(**{**O.config_c4_small().to_kwargs(), "use_scale_shift_norm": False})
        net, w = make(cfg)
        net = net.cuda()
        x, t = T(g["x"]), T(g["t"])
        y = net(x.cuda(), t.cuda()).cpu()
        assert rel(y, T(g["y"])) < FP32_TIGHT
        hd = net.native(torch.device("cuda", torch.cuda.current_device()))
        names = [k[4:] for k in g.files if k.startswith("tap_")]
        assert len(names) == 9
        for k in names:
            got = hd.tap(k, 2, torch.device("cuda")).cpu()
            assert rel(got.reshape(2, -1)[:, ::16], T(g[f"tap_{k}"])) < FP32_TIGHT, k
        return
    cfg = O.ADMConfig(**{**O.config_c4_small().to_kwargs(), "use_scale_shift_norm": False, "model_channels": 64})
    net, w = make(cfg, "bf16", seed=5)
    gen = torch.Generator().manual_seed(21)
    x, t = torch.randn(2, cfg.in_channels, 16, 32, generator=gen), torch.tensor([-0.9, 0.4])
    net = net.cuda()
    y = net(x.cuda(), t.cuda()).cpu()
    hd = net.native(torch.device("cuda", torch.cuda.current_device()))
    taps = {k: hd.tap(k, 2, torch.device("cuda")).cpu() for k in hd.tap_names()}
    errs = {}
    with torch.no_grad():
        y_f = O.unet2d_forward(w, cfg, x, t, storage="bf16", force=taps, errs=errs)
    assert set(errs) == set(taps) and any(k.endswith(".h1") for k in errs)
    for k, e in errs.items():
        assert e < (BF16_ATT_TOL if k.endswith(".att") else BF16_CONV_TOL), (k, e)
    assert O.rel_l2(y, y_f) < BF16_CONV_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["updown", "pool"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_resampling_variants_vs_reference_golden_and_oracle(gold, tag, dtype):
    """The other resampling forms of the constructor (VERDICT r2 "missing" 2): ``resblock_updown=True`` (ResBlock(up / down=True): avg-pooled or
    nearest-upsampled activation AND input, :197-207, :249-254; here together with additive conditioning, as the reference fixture has it) and
    ``conv_resample=False`` (AvgPool2d / nearest interpolation between the levels, :122-125, :153-156).  fp32 against the reference's own
    outputs (fixtures of oracle/gen_golden_next.py); bf16 per stored tensor against the bf16-storage oracle (64-channel widths)."""
    kw = O.config_c4_small().to_kwargs()
    var = {"updown": {"resblock_updown": True, "use_scale_shift_norm": False},
           "pool": {"conv_resample": False, "attention_resolutions": "32,16", "channel_mult": (1, 1, 2)}}[tag]
    if dtype == "fp32":
        cfg = O.ADMConfig(**{**kw, **var})
        net, w = make(cfg)
        net = net.cuda()
        if tag == "pool":
            # three levels: the fixture's 16 x 32 input leaves 32 pixels at the coarsest one, below the device's 64-pixel tile -- the oracle (pinned
            # to the reference on that fixture by tests/test_oracle_next.py) is the checker here, at 32 x 64
            x, t = torch.randn(2, cfg.in_channels, 32, 64, generator=torch.Generator().manual_seed(41)), torch.tensor([-0.9, 0.4])
            taps_o = {}
            with torch.no_grad():
                y_o = O.unet2d_forward(w, cfg, x, t, taps=taps_o)
            y = net(x.cuda(), t.cuda()).cpu()
            assert rel(y, y_o) < FP32_TIGHT
            hd = net.native(torch.device("cuda", torch.cuda.current_device()))
            names = [k for k in hd.tap_names() if k in taps_o]
            assert len(names) >= 9
            for k in names:
                assert rel(hd.tap(k, 2, torch.device("cuda")).cpu().reshape(2, -1), taps_o[k].reshape(2, -1)) < FP32_TIGHT, k
            return
        x, t = T(gold[f"adm_{tag}_x"]), T(gold[f"adm_{tag}_t"])
        y = net(x.cuda(), t.cuda()).cpu()
        assert rel(y, T(gold[f"adm_{tag}_y"])) < FP32_TIGHT
        hd = net.native(torch.device("cuda", torch.cuda.current_device()))
        names = [k[len(f"adm_{tag}_tap_"):] for k in gold.files if k.startswith(f"adm_{tag}_tap_")]
        assert len(names) >= 9
        for k in names:
            got = hd.tap(k, x.shape[0], torch.device("cuda")).cpu()
            assert rel(got.reshape(x.shape[0], -1)[:, ::16], T(gold[f"adm_{tag}_tap_{k}"])) < FP32_TIGHT, k
        return
    cfg = O.ADMConfig(**{**kw, **var, "model_channels": 64})
    net, w = make(cfg, "bf16", seed=6)
    gen = torch.Generator().manual_seed(31)
    x, t = torch.randn(2, cfg.in_channels, 32 if tag == "pool" else 16, 64 if tag == "pool" else 32, generator=gen), torch.tensor([-0.9, 0.4])
    net = net.cuda()
    y = net(x.cuda(), t.cuda()).cpu()
    hd = net.native(torch.device("cuda", torch.cuda.current_device()))
    taps = {k: hd.tap(k, 2, torch.device("cuda")).cpu() for k in hd.tap_names()}
    errs = {}
    with torch.no_grad():
        y_f = O.unet2d_forward(w, cfg, x, t, storage="bf16", force=taps, errs=errs)
    assert set(errs) == set(taps)
    for k, e in errs.items():
        assert e < (BF16_ATT_TOL if k.endswith(".att") else BF16_CONV_TOL), (k, e)
    assert O.rel_l2(y, y_f) < BF16_CONV_TOL


@pytest.mark.gpu
def test_config4_sampler_with_injected_draws_vs_reference_golden(gold):
    """EDMSampler(s_churn=40, s_noise=1.003, s_tmin=0.05, s_tmax=50, num_steps=35): 69 evaluations on a 4-D state, the
    reference's randn_like draws injected; eager and graph-replayed; the denoise wrapper at one sigma."""
    cfg = A.config_c4_small()
    net, w = make(cfg)
    net = net.cuda()
    diff = A.EluDiffusion(sigma_data=0.5)
    noise, draws, sig = T(gold["adm_samp_noise"]), T(gold["adm_samp_draws"]), T(gold["adm_samp_sigmas"])
    for use_graph in (False, True, True):
        smp = A.EDMSampler(s_tmin=0.05, s_tmax=50.0, s_churn=40.0, s_noise=1.003, num_steps=35, use_graph=use_graph)
        y = smp(noise.cuda(), fn=diff.denoise_fn, net=net, sigmas=sig, injected_noise=draws.cuda()).cpu()
        assert y.shape == noise.shape
        assert rel(y, T(gold["adm_samp_y"])) < 2e-4, (use_graph, rel(y, T(gold["adm_samp_y"])))
    from oracle import edm as E
    with torch.no_grad():
        d = diff.denoise_fn(noise.cuda(), net=net, inference=True, sigma=1.7).cpu()
        ref = E.denoise(lambda xi, ti, **kw: O.unet2d_forward(w, cfg, xi, ti), noise, 0.5, sigma=1.7)
    assert rel(d, ref) < FP32_TIGHT


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_config4_exactly_full_net_35_step_churn_sampler_vs_oracle():
    """BASELINE configs[3] at full size: the 71 M-parameter `config_c4()` net on one 1 x 80 x 256 mel block,
    EDMSampler(s_churn=40, s_noise=1.003, s_tmin=0.05, s_tmax=50, num_steps=35) = 69 evaluations with injected draws, fp32 mode,
    eager and graph-replayed, against the oracle's loop around the pinned network restatement (sampler_edm.py:333-397)."""
    from oracle import edm as E, samplers as S
    cfg = A.config_c4()
    net, w = make(cfg, seed=4)
    net = net.cuda()
    diff = A.EluDiffusion(sigma_data=0.5)
    g = torch.Generator().manual_seed(41)
    noise = torch.randn(1, 1, 80, 256, generator=g)
    draws = torch.randn(35, 1, 1, 80, 256, generator=g)
    sig = A.KarrasSchedule(0.002, 80.0, 7.0, 35)()
    fn_o = lambda xx, sigma=None, sigmas=None: E.denoise(lambda xi, ti, **kw: O.unet2d_forward(w, cfg, xi, ti), xx, 0.5, sigma=sigma, sigmas=sigmas)
    with torch.no_grad():
        ref = S.edm_sampler(noise, fn_o, sig, 35, s_tmin=0.05, s_tmax=50.0, s_churn=40.0, s_noise=1.003, injected_noise=draws)
    assert float(ref.abs().max()) > 1e-3
    for use_graph in (False, True):
        smp = A.EDMSampler(s_tmin=0.05, s_tmax=50.0, s_churn=40.0, s_noise=1.003, num_steps=35, use_graph=use_graph)
        y = smp(noise.cuda(), fn=diff.denoise_fn, net=net, sigmas=sig, injected_noise=draws.cuda()).cpu()
        assert rel(y, ref) < 2e-4, (use_graph, rel(y, ref))           # 69 chained evaluations; north-star bar 1e-3


@pytest.mark.gpu
def test_two_image_shapes_with_equal_area_do_not_share_a_captured_graph():
    """[B,1,16,64] and [B,1,32,32] have the same H * W; a graph captured for the first shape must not be
    replayed for the second (the conv2d geometry is baked into the captured launches)."""
    from oracle import edm as E, samplers as S
    cfg = A.config_c4_small()
    net, w = make(cfg)
    net = net.cuda()
    diff = A.EluDiffusion(sigma_data=0.5)
    sig = A.KarrasSchedule(0.002, 80.0, 7.0, 4)()
    g = torch.Generator().manual_seed(8)
    fn_o = lambda xx, sigma=None, sigmas=None: E.denoise(lambda xi, ti, **kw: O.unet2d_forward(w, cfg, xi, ti), xx, 0.5, sigma=sigma, sigmas=sigmas)
    smp = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=4, use_graph=True)
    for shape in ((1, cfg.in_channels, 16, 64), (1, cfg.in_channels, 32, 32), (1, cfg.in_channels, 16, 64)):
        noise = torch.randn(*shape, generator=g)
        with torch.no_grad():
            ref = S.edm_sampler(noise, fn_o, sig, 4, s_churn=0.0, s_noise=1.0)
        y = smp(noise.cuda(), fn=diff.denoise_fn, net=net, sigmas=sig).cpu()
        assert rel(y, ref) < FP32_TIGHT, (shape, rel(y, ref))
        hd = net.native(torch.device("cuda", torch.cuda.current_device()))
        y1 = net(noise.cuda(), torch.tensor([0.1]).cuda()).cpu()      # taps of the last pass carry this shape's geometry
        with torch.no_grad():
            assert rel(y1, O.unet2d_forward(w, cfg, noise, torch.tensor([0.1]))) < FP32_TIGHT


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["o2_log", "o3_log", "o1_log", "o2_sig", "o3_sig", "o2_log_eps"])
def test_unipc_sampler_on_the_device_vs_reference_golden(gold, tag):
    """UniPCSampler (4-D states only in the reference) on the device, eager and graph-replayed, against the REFERENCE's results."""
    cases = {"o2_log": dict(order=2, log_time_spacing=True), "o3_log": dict(order=3, log_time_spacing=True),
             "o1_log": dict(order=1, log_time_spacing=True), "o2_sig": dict(order=2, log_time_spacing=False),
             "o3_sig": dict(order=3, log_time_spacing=False), "o2_log_eps": dict(order=2, log_time_spacing=True, x0_pred=False)}
    cfg = A.config_c4_small()
    net, _ = make(cfg)
    net = net.cuda()
    diff = A.EluDiffusion(sigma_data=0.5)
    noise, sig = T(gold["adm_samp_noise"]), T(gold["unipc_sigmas"])
    for use_graph in (False, True):
        smp = A.UniPCSampler(num_steps=10, use_graph=use_graph, **cases[tag])
        y = smp(noise.cuda(), fn=diff.denoise_fn, net=net, sigmas=sig).cpu()
        assert rel(y, T(gold[f"unipc_{tag}_y"])) < 2e-4, (tag, use_graph, rel(y, T(gold[f"unipc_{tag}_y"])))


@pytest.mark.gpu
def test_unipc_sampler_on_a_1d_waveform_state_vs_oracle():
    """The device UniPC is shape-agnostic (the reference's is not: its einsum wants [B, K, C, H, W]): the 1-D U-Net, 12 steps, order 3."""
    from audiodiffuser_amd.weights import generate_weights as gw, generate_noise
    from oracle import edm as E, samplers as S
    cfg = A.config_tiny()
    w = gw(cfg, seed=0)
    net = A.UNet1dBase.from_config(cfg, compute_dtype="fp32")
    net.load_state_dict(w)
    noise = generate_noise(3, 2, 256)
    sig = A.KarrasSchedule(0.002, 80.0, 7.0, 12)()
    with torch.no_grad():
        ref = S.unipc_sampler(noise, E.make_denoiser(w, cfg, 0.2), sig, 12, order=3, log_time_spacing=True)
    y = A.UniPCSampler(num_steps=12, order=3)(noise.cuda(), fn=A.EluDiffusion(sigma_data=0.2).denoise_fn, net=net.cuda(), sigmas=sig).cpu()
    assert rel(y, ref) < 2e-4, rel(y, ref)


def _cls_cfg():
    return A.ADMConfig(**{**A.config_c4_small().to_kwargs(), "num_classes": 5, "use_new_attention_order": True, "num_head_channels": 16})


@pytest.mark.gpu
def test_class_conditional_forward_vs_reference_golden(gold):
    """UNetModel(num_classes=...) (the shipped diffunet_complex_oai_sc09_cfg.yaml setting): labels kept (cond_drop_prob 0) and dropped (1) against
    the REFERENCE's outputs (fixture 'cls_new': class-conditional + new attention order + per-head width 16)."""
    cfg = _cls_cfg()
    net, _ = make(cfg)
    net = net.cuda()
    x, t, cl = T(gold["adm_cls_new_x"]), T(gold["adm_cls_new_t"]), T(gold["adm_cls_new_classes"])
    y = net(x.cuda(), t.cuda(), classes=cl.cuda(), cond_drop_prob=0.0).cpu()
    assert rel(y, T(gold["adm_cls_new_y"])) < FP32_TIGHT
    y0 = net(x.cuda(), t.cuda(), classes=cl.cuda(), cond_drop_prob=1.0).cpu()
    assert rel(y0, T(gold["adm_cls_new_y_null"])) < FP32_TIGHT
    with pytest.raises(AssertionError):
        net(x.cuda(), t.cuda())
    with pytest.raises(IndexError):
        net(x.cuda(), t.cuda(), classes=torch.tensor([1, 9]).cuda())


@pytest.mark.gpu
def test_class_conditional_guidance_denoise_and_adpm2_sampler_vs_oracle():
    """Classifier-free guidance (cond_scale 4, the shipped value) through denoise_fn and the module's default sampler (ADPM2, ancestral noise injected),
    eager and graph-replayed, against the oracle's wrapper + loop."""
    from oracle import edm as E, samplers as S
    cfg = _cls_cfg()
    net, w = make(cfg)
    net = net.cuda()
    diff = A.EluDiffusion(sigma_data=0.2)
    g = torch.Generator().manual_seed(31)
    x, cl = torch.randn(2, 1, 16, 32, generator=g), torch.tensor([3, 0])

    def net_o(xi, ti, cond_drop_prob=0.0):
        return O.unet2d_forward(w, cfg, xi, ti, classes=cl, cond_drop_prob=cond_drop_prob)

    fn_o = lambda xx, sigma=None, sigmas=None: E.denoise(net_o, xx, 0.2, sigma=sigma, sigmas=sigmas, cond_scale=4.0)
    with torch.no_grad():
        d = diff.denoise_fn(x.cuda() * 2.0, net=net, inference=True, cond_scale=4.0, sigma=2.0, classes=cl.cuda()).cpu()
        assert rel(d, fn_o(x * 2.0, sigma=2.0)) < FP32_TIGHT
        sig = A.KarrasSchedule(0.001, 30.0, 9.0, 8)()
        draws = torch.randn(7, 2, 1, 16, 32, generator=g)
        ref = S.adpm2_sampler(x, fn_o, sig, 8, rho=1.0, injected_noise=draws)
        for use_graph in (False, True):
            smp = A.ADPM2Sampler(rho=1.0, num_steps=8, cond_scale=4.0, use_graph=use_graph)
            y = smp(x.cuda(), fn=diff.denoise_fn, net=net, sigmas=sig, classes=cl.cuda(), injected_noise=draws.cuda()).cpu()
            assert rel(y, ref) < 2e-4, (use_graph, rel(y, ref))


@pytest.mark.gpu
def test_two_channel_complex_stft_layout_vs_oracle():
    """The shipped pipeline's tensor layout (diffunet_complex_oai_sc09_cfg.yaml: real / imaginary STFT planes = 2 input and 2 output channels, 10
    classes, guidance) on the small net at 2 x 32 x 64: the first / last conv with more than one channel, labels, a guided denoise."""
    from oracle import edm as E
    cfg = A.ADMConfig(**{**A.config_c4_small().to_kwargs(), "in_channels": 2, "out_channels": 2, "num_classes": 10})
    net, w = make(cfg, seed=13)
    net = net.cuda()
    g = torch.Generator().manual_seed(14)
    x, t, cl = torch.randn(3, 2, 32, 64, generator=g), torch.tensor([0.4, -0.9, 0.0]), torch.tensor([9, 0, 4])
    with torch.no_grad():
        ref = O.unet2d_forward(w, cfg, x, t, classes=cl)
    y = net(x.cuda(), t.cuda(), classes=cl.cuda()).cpu()
    assert y.shape == x.shape and rel(y, ref) < FP32_TIGHT
    diff = A.EluDiffusion(sigma_data=0.2)
    net_o = lambda xi, ti, cond_drop_prob=0.0: O.unet2d_forward(w, cfg, xi, ti, classes=cl, cond_drop_prob=cond_drop_prob)
    with torch.no_grad():
        d = diff.denoise_fn(x.cuda(), net=net, inference=True, cond_scale=4.0, sigmas=torch.tensor([0.5, 3.0, 20.0]).cuda(), classes=cl.cuda()).cpu()
        refd = E.denoise(net_o, x, 0.2, sigmas=torch.tensor([0.5, 3.0, 20.0]), cond_scale=4.0)
    assert rel(d, refd) < FP32_TIGHT
