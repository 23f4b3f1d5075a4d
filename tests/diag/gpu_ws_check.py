"""Compare the warp-specialised GEMM path against the oracle at a size where it is selected (tiles >= 256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_noise
from gpu_helpers import tap_errors
cfg = A.config_c2()
B = int(os.environ.get("B", "8"))
x = generate_noise(0, B, 16384) * 0.7
t = torch.linspace(-1.0, 0.5, B)
for dtype in ("fp32", "bf16"):
    errs, y, yo = tap_errors(cfg, x, t, dtype, 0)
    worst = max(errs, key=errs.get)
    print(dtype, "WS=" + os.environ.get("ADF_GEMM_WS", "1"), "out", f"{errs['out']:.3e}", "worst", worst, f"{errs[worst]:.3e}", flush=True)
