"""Diagnostic (GPU box): per-layer deviation of the HIP bf16 path from the bf16-storage oracle, teacher-forced and chained,
for the cases the parity tests assert on.  Writes gpurun_out/bf16_parity_report.json; the tolerances in
tests/test_gpu_parity.py were set from this report (profiles/r02_bf16_parity_report.json is a committed copy)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import audiodiffuser_amd as A  # noqa: E402
from audiodiffuser_amd.weights import generate_noise  # noqa: E402
from gpu_helpers import tap_errors_bf16, golden_inputs  # noqa: E402

cases = {
    "tiny_B2_L256": (A.config_tiny(), *golden_inputs("tiny")),
    "c1_B2_L2048": (A.config_c1(), *golden_inputs("c1")),
    "c3_B2_L4096": (A.config_c3(), generate_noise(0, 2, 4096) * 0.7, torch.tensor([-0.9, 0.35])),
    "c2_B8_L16384": (A.config_c2(), generate_noise(0, 8, 16384) * 0.7, torch.linspace(-1.0, 0.5, 8)),
    "c3_B1_L16384": (A.config_c3(), generate_noise(3, 1, 16384) * 0.6, torch.tensor([0.2])),
}
if len(sys.argv) > 1 and sys.argv[1] == "full":
    cases["c2_B64_L16384"] = (A.config_c2(), generate_noise(0, 64, 16384) * 0.7, torch.linspace(-1.2, 0.6, 64))
rep = {}
for name, (cfg, x, t) in cases.items():
    t0 = time.time()
    forced, chain, y, yf, yc = tap_errors_bf16(cfg, x, t)
    rep[name] = {"forced_max": max(forced.values()), "chain_max": max(chain.values()), "forced": forced, "chain": chain,
                 "seconds": time.time() - t0}
    print(name, "forced max %.3e (%s)" % (rep[name]["forced_max"], max(forced, key=forced.get)),
          "chain max %.3e (%s)" % (rep[name]["chain_max"], max(chain, key=chain.get)), "%.1fs" % (time.time() - t0), flush=True)
# ---- the benched mode through the sampler -------------------------------------------------------------------------
from oracle import edm as E, samplers as S  # noqa: E402
from gpu_helpers import make_net, rel_l2, rel_err  # noqa: E402
cfg = A.config_c2()
net16, w = make_net(cfg, "bf16")
net32, _ = make_net(cfg, "fp32")
d = A.EluDiffusion(sigma_data=0.2)
noise = generate_noise(777, 2, 16384)
sig8 = A.KarrasSchedule(0.002, 80.0, 7.0, 8)()
t0 = time.time()
y16 = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=8)(noise.cuda(), fn=d.denoise_fn, net=net16, sigmas=sig8).cpu()
y32 = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=8)(noise.cuda(), fn=d.denoise_fn, net=net32, sigmas=sig8).cpu()
with torch.no_grad():
    yo = S.edm_sampler(noise, E.make_denoiser(w, cfg, 0.2, storage="bf16"), sig8, 8, s_churn=0.0, s_noise=1.0)
rep["heun8_c2_B2"] = {"bf16_dev_vs_bf16_oracle_rel_l2": rel_l2(y16, yo), "bf16_dev_vs_fp32_dev_rel_l2": rel_l2(y16, y32),
                      "bf16_oracle_vs_fp32_dev_rel_l2": rel_l2(yo, y32), "seconds": time.time() - t0}
print("heun8", rep["heun8_c2_B2"], flush=True)
sig50 = A.KarrasSchedule(0.002, 80.0, 7.0, 50)()
smp = A.EDMSampler(s_churn=0.0, s_noise=1.0, num_steps=50)
a = smp(noise.cuda(), fn=d.denoise_fn, net=net16, sigmas=sig50).cpu()
b = smp(noise.cuda(), fn=d.denoise_fn, net=net32, sigmas=sig50).cpu()
rep["heun50_c2_B2"] = {"bf16_vs_fp32_rel_l2": rel_l2(a, b), "bf16_vs_fp32_max_over_max": rel_err(a, b)}
print("heun50", rep["heun50_c2_B2"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "bf16_parity_report.json"), "w"), indent=1)
