"""CPU: host-side logic of the plugin surface and the C-ABI library's loadability.
No compute call is made on the library here (there is no GPU in the build container)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

import audiodiffuser_amd as A
from audiodiffuser_amd import _lib
from audiodiffuser_amd.distributed import shard_range, rank_noise
from audiodiffuser_amd.weights import generate_noise, generate_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# ---- C ABI ----------------------------------------------------------------------------------------
def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "audiodiffuser_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(adf_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load_library()                  # raises if the .so is missing: no fallback
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/audiodiffuser_amd.h but not exported"
    assert set(syms) == set(_lib.EXPORTS), "ctypes binding table and header disagree"


def test_abi_version_agrees_across_header_binding_library_and_docs():
    """INTEGRATION.md asserted version 3 while header and binding said 4 -- the documented binding refused the shipped library."""
    hdr = open(os.path.join(ROOT, "include", "audiodiffuser_amd.h")).read()
    ver = int(re.search(r"#define ADF_ABI_VERSION (\d+)", hdr).group(1))
    assert ver == _lib.ABI_VERSION == _lib.load_library().adf_abi_version()
    # the header's changelog has a line for every version since 3
    for v in range(3, ver + 1):
        assert re.search(rf"^ \* {v}: ", hdr, flags=re.M), f"no changelog entry '{v}:' in the header comment"
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    # any literal the docs compare adf_abi_version() with must be the current one; the example reads it from the header instead
    for lit in re.findall(r"adf_abi_version\(\)\s*==\s*(\d+)", doc):
        assert int(lit) == ver, f"INTEGRATION.md asserts ABI version {lit}, the header says {ver}"
    assert "adf_abi_version() == ADF_ABI_VERSION" in doc
    for lit in re.findall(r"ADF_ABI_VERSION[^\n]*#\s*(\d+) in this tree", doc):
        assert int(lit) == ver
    assert _lib.DTYPE_F32X3 == int(re.search(r"#define ADF_DTYPE_F32X3 (\d+)", hdr).group(1))


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.AdfNetConfig) == 4 * (9 + 13 + 12 * 3 + 4 + 2 + 1)
    assert C.sizeof(_lib.AdfSamplerDesc) == 4 * 16


def test_sampler_nfe_via_abi():
    """adf_sampler_nfe walks the same host-side step logic as adf_sampler_run without touching the GPU."""
    lib = _lib.load_library()
    def nfe(desc, n):
        sg = A.KarrasSchedule(0.002, 80.0, 7.0, n)()
        arr = (C.c_float * n)(*sg.tolist())
        return lib.adf_sampler_nfe(C.byref(desc), arr, n)
    assert nfe(A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=18)._desc(0.2), 18) == 35
    assert nfe(A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=50)._desc(0.2), 50) == 99
    assert nfe(A.EDMSampler(s_tmin=0.05, s_tmax=50.0, s_churn=40.0, s_noise=1.003, num_steps=35)._desc(0.2), 35) == 69
    assert nfe(A.EDMSampler(s_churn=0.0, num_steps=18, use_heun=False)._desc(0.2), 18) == 18
    assert nfe(A.EDMAlphaSampler(alpha=1.0, num_steps=18)._desc(0.2), 18) == 34
    assert nfe(A.DPMSampler(1.0, order=3, num_steps=50, multisteps=True, x0_pred=True, log_time_spacing=False)._desc(0.2), 50) == 49
    bad = A.DPMSampler(1.0, order=4, num_steps=50, multisteps=True, x0_pred=True, log_time_spacing=False)._desc(0.2)
    assert nfe(bad, 50) == -1
    assert nfe(A.DPM2Sampler(num_steps=50, s_churn=0.0)._desc(0.2), 50) == 2 * 49      # no sigma_next == 0 inside the schedule
    assert nfe(A.UniPCSampler(num_steps=20, order=2)._desc(0.2), 20) == 20            # "NFE = num_steps" (sampler_edm.py:813)
    assert nfe(A.UniPCSampler(num_steps=20, order=3, log_time_spacing=False)._desc(0.2), 20) == 19
    assert nfe(A.UniPCSampler(num_steps=20, order=4)._desc(0.2), 20) == -1
    assert nfe(A.ADPM2Sampler(num_steps=50)._desc(0.2), 50) == 2 * 49
    assert nfe(A.ADPMPP2SSampler(num_steps=50)._desc(0.2), 50) == 2 * 49            # sigma_down > 0 on a Karras schedule: two evaluations per step
    assert nfe(A.DPM2MSampler(num_steps=50, reflow=True)._desc(0.2), 51) == 50


# ---- plugin surface -------------------------------------------------------------------------------
def test_karras_schedule_plugin(golden):
    for n in (18, 35, 50):
        s = A.KarrasSchedule(sigma_min=0.002, sigma_max=80.0, rho=7.0, num_steps=n)()
        assert s.dtype == torch.float32 and s.device.type == "cpu"
        assert torch.equal(s, T(golden[f"karras_{n}"]))


def test_net_plugin_state_dict_contract():
    lay = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))
    for tag, cfg in (("tiny", A.config_tiny()), ("c1", A.config_c1())):
        net = A.UNet1dBase(**cfg.to_kwargs())          # hydra-style kwargs construction
        sd = net.state_dict()
        assert list(sd.keys()) == list(lay[tag]["keys"].keys())
        assert all(list(v.shape) == lay[tag]["keys"][k] for k, v in sd.items())
        assert sum(p.numel() for p in net.parameters()) == lay[tag]["num_params"]
        assert next(net.parameters()).dtype == torch.float32           # dtype probe of the Lightning module
        assert float(sd["unet.to_out.to_out.weight"].abs().max()) == 0.0  # zero-init output layer (unet1d.py:619)
        net.load_state_dict(generate_weights(cfg), strict=True)         # strict load of reference-keyed tensors


def test_net_plugin_refuses_cpu_and_conditioning():
    net = A.UNet1dBase.from_config(A.config_tiny())
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 64), torch.zeros(1))      # no CPU fallback for the HIP path
    with pytest.raises(NotImplementedError):
        A.UNet1dBase(text_cond=True, **A.config_tiny().to_kwargs())
    with pytest.raises(NotImplementedError):
        A.UNet1dBase(class_cond=True, class_embed_dim=32, **A.config_tiny().to_kwargs())
    with pytest.raises(ValueError):
        A.UNet1dBase(class_cond=True, **A.config_tiny().to_kwargs())      # needs num_classes
    with pytest.raises(ValueError):
        A.UNet1dBase(compute_dtype="fp8", **A.config_tiny().to_kwargs())


def test_class_cond_plugin_contract():
    """class_cond=True: LabelEmbedder parameters come first and every FiLM Linear reads cat(time, class) embeddings,
    exactly the reference's state_dict (tests/golden/state_dict_layout.json "tiny_cc")."""
    import json
    lay = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))["tiny_cc"]
    net = A.UNet1dBase.from_config(A.config_tiny_cc())
    sd = net.state_dict()
    assert list(sd.keys()) == list(lay["keys"].keys())
    assert all(list(v.shape) == lay["keys"][k] for k, v in sd.items())
    c = _lib.make_config(net.cfg, _lib.DTYPE_F32)
    assert c.num_classes == 10 and _lib.make_config(A.config_tiny(), _lib.DTYPE_F32).num_classes == 0
    # the fast path takes `classes` (and any cond_scale) only for a class-conditional net
    d = A.EluDiffusion(sigma_data=0.2)
    assert d._native_ok(net, True, 3.0, {"classes": torch.zeros(2, dtype=torch.int64)})
    assert not d._native_ok(net, True, 3.0, {"classes": torch.zeros(2, dtype=torch.int64), "text_embeds": torch.zeros(1)})
    plain = A.UNet1dBase.from_config(A.config_tiny())
    assert d._native_ok(plain, True, 1.0, {}) and not d._native_ok(plain, True, 2.0, {})


def test_scale_weights_plugin(golden):
    d = A.EluDiffusion(sigma_data=0.2)
    c_skip, c_out, c_in, c_noise = d.get_scale_weights(T(golden["scale_sigmas"]), 3)
    for name, v in (("c_skip", c_skip), ("c_out", c_out), ("c_in", c_in), ("c_noise", c_noise)):
        assert torch.equal(v.reshape(-1), T(golden[f"scale_{name}"]))


def test_denoise_fn_compat_branch_matches_oracle(golden):
    """Foreign net callable -> tensor-op branch (interface compatibility), same math as the oracle."""
    from oracle import unet1d as O
    cfg = A.config_tiny()
    w = generate_weights(cfg)
    net = lambda x, t, **kw: O.unet1d_forward(w, cfg, x, t)
    d = A.EluDiffusion(sigma_data=0.2)
    xn = generate_noise(7, 2, 256)
    with torch.no_grad():
        y = d.denoise_fn(xn * 1.5, net=net, sigma=torch.tensor(1.5), inference=True, cond_scale=1.0)
    assert rel(y, T(golden["denoise_tiny_1"])) < 1e-5
    with pytest.raises(AssertionError):
        d.denoise_fn(xn, net=net, inference=True)


def test_samplers_compat_branch_mock(golden):
    mock = lambda x, net=None, sigma=None, **kw: 0.5 * x
    noise = generate_noise(40, 2, 256)
    s18, s50 = T(golden["karras_18"]), T(golden["karras_50"])
    y = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=18)(noise, fn=mock, net=None, sigmas=s18)
    assert rel(y, T(golden["smp_heun18_tiny_mock_final"])) < 1e-6
    y = A.EDMAlphaSampler(alpha=1.0, num_steps=18)(noise, fn=mock, net=None, sigmas=s18)
    assert rel(y, T(golden["smp_alpha18_tiny_mock_final"])) < 1e-6
    y = A.DPMSampler(1.0, order=3, num_steps=50, multisteps=True, x0_pred=True, log_time_spacing=False)(noise, fn=mock, net=None, sigmas=s50)
    assert rel(y, T(golden["smp_dpm50_tiny_mock_final"])) < 1e-6
    with pytest.raises(NotImplementedError):
        A.DPMSampler(1.0, order=3, num_steps=50, multisteps=False, x0_pred=False)(noise, fn=mock, net=None, sigmas=s50)   # foreign fn: native path only


def test_lms_and_dpm_variants_compat_branch(golden):
    """The tensor-op branches of LMSSampler / DPMSampler (foreign denoiser callable) against the reference's results."""
    from oracle import edm as E
    cfg = A.config_tiny()
    den = E.make_denoiser(generate_weights(cfg, seed=0), cfg, 0.2)
    fn = lambda x, net=None, sigma=None, **kw: den(x, sigma=sigma)
    noise = generate_noise(70, 2, 256)
    sig = A.KarrasSchedule(0.002, 80.0, 7.0, 10)()
    with torch.no_grad():
        for order in (4, 2):
            y = A.LMSSampler(num_steps=10, order=order)(noise, fn=fn, net=None, sigmas=sig)
            assert rel(y, T(golden[f"smp_lms10_o{order}_final"])) < 5e-4
        for order, logsp, n in ((3, True, 10), (3, True, 9), (2, True, 7), (1, True, 4), (3, False, 10), (2, False, 10)):
            tag = f"o{order}_{'log' if logsp else 'lin'}_n{n}"
            smp = A.DPMSampler(1.0, order=order, num_steps=n, multisteps=False, log_time_spacing=logsp)
            y = smp(noise, fn=fn, net=None, sigmas=A.KarrasSchedule(0.002, 80.0, 7.0, n)())
            assert rel(y, T(golden[f"smp_dpm_single_{tag}_final"])) < 5e-4, tag
        for order in (3, 2):
            y = A.DPMSampler(1.0, order=order, num_steps=10, multisteps=True, log_time_spacing=True)(noise, fn=fn, net=None, sigmas=sig)
            assert rel(y, T(golden[f"smp_dpm_multi_log_o{order}_final"])) < 5e-4
        for tag, sg2m in (("k11", A.KarrasSchedule(0.002, 80.0, 7.0, 11)()), ("k10_zero", torch.cat([sig, torch.zeros(1)]))):
            y = A.DPM2MSampler(num_steps=10)(noise, fn=fn, net=None, sigmas=sg2m)
            assert rel(y, T(golden[f"smp_dpm2m_{tag}_final"])) < 5e-4, tag
        with pytest.raises(IndexError):
            A.DPM2MSampler(num_steps=10)(noise, fn=fn, net=None, sigmas=sig)


def test_lms_coefficients_match_scipy_quad():
    """The exact Gauss-Legendre integrals against the reference's scipy quad call (sampler_edm.py:1149-1160), which runs
    on fp32 NumPy scalars (under NumPy 2 promotion the whole basis polynomial is evaluated in fp32): agreement to fp32 noise."""
    from oracle import samplers as S
    t = A.KarrasSchedule(0.002, 80.0, 7.0, 12)().numpy()
    for i in range(11):
        cur = min(i + 1, 4)
        for j in range(cur):
            assert abs(A.LMSSampler.linear_multistep_coeff(cur, t, i, j) - S.lms_coeff(cur, t, i, j)) <= 5e-7 * max(1.0, abs(S.lms_coeff(cur, t, i, j)))
    with pytest.raises(ValueError):
        A.LMSSampler.linear_multistep_coeff(3, t, 1, 0)


def test_dpm_nfe_counts():
    assert A.DPMSampler(1.0, order=3, num_steps=50, multisteps=True, log_time_spacing=False).nfe() == 49
    assert A.DPMSampler(1.0, order=3, num_steps=10, multisteps=False, log_time_spacing=True).nfe() == 10
    assert A.DPMSampler(1.0, order=3, num_steps=10, multisteps=False, log_time_spacing=False).nfe() == 9
    assert A.DPMSampler(1.0, order=2, num_steps=7, multisteps=False, log_time_spacing=True).nfe() == 7


def test_sampler_kwargs_forwarded_to_fn():
    seen = {}
    def fn(x, net=None, sigma=None, inference=None, cond_scale=None, **kw):
        seen.update(kw)
        seen["cond_scale"] = cond_scale
        return 0.5 * x
    A.EDMSampler(s_churn=0.0, num_steps=3, cond_scale=2.5)(torch.zeros(1, 1, 8), fn=fn, net=None,
                                                            sigmas=A.KarrasSchedule(0.002, 80.0, 7.0, 3)(), classes=torch.tensor([3]))
    assert "classes" in seen and seen["cond_scale"] == 2.5


# ---- sharding helpers -----------------------------------------------------------------------------
def test_shard_ranges_partition_the_batch():
    for gb in (1, 7, 64, 512, 1000):
        for world in (1, 2, 3, 8):
            r = [shard_range(gb, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == gb
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_rank_noise_is_keyed_by_global_index():
    full = generate_noise(0, 8, 64)
    parts = torch.cat([rank_noise(8, 64, r, 4) for r in range(4)])
    assert torch.equal(full, parts)


def test_dpm2_family_compat_branch_matches_oracle():
    """The tensor-op branch of DPM2Sampler / ADPM2Sampler (foreign fn / net) against the oracle loops, mock denoiser."""
    from oracle import samplers as S
    from audiodiffuser_amd.weights import generate_noise
    noise = generate_noise(3, 2, 128)
    sig = A.KarrasSchedule(0.002, 80.0, 7.0, 9)()
    inj = torch.stack([generate_noise(100 + i, 2, 128) for i in range(8)])
    mock = lambda x, net=None, sigma=None, **k: 0.5 * x / (1 + sigma)
    fn_o = lambda x, sigma=None: 0.5 * x / (1 + sigma)
    y = A.DPM2Sampler(num_steps=9, s_tmin=0.05, s_tmax=50.0, s_churn=20.0, s_noise=1.01)(noise, fn=mock, net=None, sigmas=sig, injected_noise=inj)
    assert torch.equal(y, S.dpm2_sampler(noise, fn_o, sig, 9, s_tmin=0.05, s_tmax=50.0, s_churn=20.0, s_noise=1.01, injected_noise=inj))
    y = A.ADPM2Sampler(rho=7.0, num_steps=9, eta=0.8)(noise, fn=mock, net=None, sigmas=sig, injected_noise=inj)
    assert torch.equal(y, S.adpm2_sampler(noise, fn_o, sig, 9, rho=7.0, eta=0.8, injected_noise=inj))
    # the rest of stochastic_sampler_edm.py: DPM++ 2S a (also on a schedule ending in 0: Euler last step, no last draw) and DPM2M with reflow
    for sg, nd in ((sig, 8), (torch.cat([sig[:8], torch.zeros(1)]), 7)):
        y = A.ADPMPP2SSampler(num_steps=9, eta=0.8)(noise, fn=mock, net=None, sigmas=sg, injected_noise=inj[:nd])
        assert torch.equal(y, S.adpmpp2s_sampler(noise, fn_o, sg, 9, eta=0.8, injected_noise=inj[:nd]))
    sg10 = A.KarrasSchedule(0.002, 80.0, 7.0, 10)()
    for reflow in (False, True):
        y = A.DPM2MSampler(num_steps=9, reflow=reflow)(noise, fn=mock, net=None, sigmas=sg10)
        assert torch.equal(y, S.dpm2m_sampler(noise, fn_o, sg10, 9, reflow=reflow))


def test_bench_defaults_per_config():
    """bench.py --config picks the BASELINE workload sizes: configs[1] 64 x 16384 / N = 50 Heun; configs[2] DPM; configs[3] 80 x 256 mel blocks,
    35-step churn sampler; configs[4] 22050 samples, 6 steps, batch 128 (1024 over 8 GPUs)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse([])
    assert (a.config, a.batch, a.length, a.num_steps, a.sampler, a.gpus) == ("c2", 64, 16384, 50, "heun", 1)
    a = bench.parse(["--config", "c3"])
    assert (a.batch, a.length, a.num_steps, a.sampler) == (64, 16384, 50, "dpm")
    a = bench.parse(["--config", "c4"])
    assert (a.batch, a.length, a.num_steps, a.sampler) == (64, 80 * 256, 35, "churn")
    a = bench.parse(["--config", "c5", "--batch", "32"])
    assert (a.batch, a.length, a.num_steps, a.sampler) == (32, 22050, 6, "heun")
    assert bench.adm_conv_flops(A.config_c4(), 1, 80, 256) > 2.0e11          # ~0.25 TFLOP per 1 x 80 x 256 block and evaluation


def test_nearest_upsample_index_map_is_integer_division():
    """``upsample_nearest_pad_kernel`` (adf_kernels.hip) takes source row ``i // f``; ``nn.Upsample(mode='nearest')`` computes
    ``floor(i * float(1 / f))``.  The two agree for every factor a UNet1dBase level can have here (2 .. 32) at lengths beyond the longest level."""
    for f in range(2, 33):
        n = 8192 if f <= 8 else 1024
        x = torch.arange(n, dtype=torch.float32)[None, None]
        y = torch.nn.Upsample(scale_factor=f, mode="nearest")(x)[0, 0].long()
        assert torch.equal(y, torch.arange(n * f) // f), f
