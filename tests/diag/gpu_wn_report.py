"""GPU diagnostic: per-layer deviation of the bf16 WaveNetNoise path from the bf16-storage oracle (teacher-forced), the fp32 path
from the oracle, and the free-running bf16 net from fp32.  Prints one JSON object (also written to gpurun_out/wn_parity_report.json).
Run as a child process by tests/test_wavenet.py with ADF_WN_WIDE=0 to cover the 64-position route of the layer kernel.
usage: python tests/diag/gpu_wn_report.py [T] [batch]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_wavenet_weights
from oracle import wavenet as W

tlen = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = A.config_c5()
w = generate_wavenet_weights(cfg, seed=5)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
audio, step = torch.randn(batch, tlen, generator=g) * 0.6, torch.linspace(-1.2, 0.5, batch)
out = {"T": tlen, "batch": batch, "route_wide": os.environ.get("ADF_WN_WIDE", "1")}
nets = {}
for dt in ("bf16", "fp32"):
    net = A.WaveNetNoise.from_config(cfg, compute_dtype=dt)
    net.load_state_dict(w)
    nets[dt] = net.to(dev)
y16 = nets["bf16"](audio.to(dev), step.to(dev)).cpu()
hd = nets["bf16"].native(dev)
taps = {n: hd.tap(n, batch, dev).cpu() for n in hd.tap_names()}
y32 = nets["fp32"](audio.to(dev), step.to(dev)).cpu()
errs = {}
with torch.no_grad():
    yf = W.wavenet_forward(w, cfg, audio, step, storage="bf16", force=taps, errs=errs)
    yo = W.wavenet_forward(w, cfg, audio, step)
worst = max(errs, key=errs.get)
out.update({"forced_max_rel_l2": errs[worst], "forced_worst_tap": worst, "forced_taps": len(errs),
            "forced_median_rel_l2": sorted(errs.values())[len(errs) // 2],
            "out_vs_forced_oracle_rel_l2": W.rel_l2(y16, yf), "bf16_vs_fp32_oracle_rel_l2": W.rel_l2(y16, yo),
            "fp32_device_vs_oracle_max_rel": float((y32 - yo).abs().max() / yo.abs().max())})
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"wn_parity_report_wide{out['route_wide']}.json"), "w") as f:
    json.dump(out, f, indent=1)
