"""Child process of tests/test_gpu_parity.py::test_unfused_launches_every_stored_tensor_vs_bf16_oracle: runs the bf16 device path with
whatever ADF_* route switches the environment holds (they are read once per process) and prints the teacher-forced relative-L2
deviation of every recorded activation from the bf16-storage oracle as one JSON line.
With a dtype argument ("f32x3", "fp32") the report is instead the free-running max-norm relative deviation of every recorded activation from the fp32 oracle.
usage: gpu_forced_report.py <preset> <batch> <length> [resnet_groups (0: the preset's)] [dtype]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import audiodiffuser_amd as A  # noqa: E402
from audiodiffuser_amd.weights import generate_noise  # noqa: E402
from gpu_helpers import tap_errors, tap_errors_bf16  # noqa: E402

preset, B, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = A.PRESETS[preset]()
if len(sys.argv) > 4 and int(sys.argv[4]) > 0:
    cfg.resnet_groups = int(sys.argv[4])
x = generate_noise(0, B, L) * 0.7
t = torch.linspace(-0.9, 0.35, B)
if len(sys.argv) > 5:
    forced, y, _ = tap_errors(cfg, x, t, sys.argv[5], 0)
else:
    forced, chain, y, yf, yc = tap_errors_bf16(cfg, x, t, chained=False)
print(json.dumps({"forced": forced, "finite": bool(torch.isfinite(y).all())}))
