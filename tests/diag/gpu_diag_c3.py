"""GPU diagnostic: per-layer error of config 3 (64 ch, attention from the 16x level) at L=4096 in bf16 and fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_noise
from gpu_helpers import tap_errors
cfg = A.config_c3()
x = generate_noise(0, 2, 4096) * 0.7
t = torch.tensor([-0.9, 0.35])
for dtype in ("fp32", "bf16"):
    errs, y, yo = tap_errors(cfg, x, t, dtype, 0)
    print(dtype, "out", f"{errs['out']:.3e}", " ".join(f"{k}={v:.1e}" for k, v in errs.items() if "attn" in k))
