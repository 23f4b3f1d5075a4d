"""GPU diagnostic (not a test): per-layer relative error of the HIP U-Net against the CPU oracle."""
import json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import audiodiffuser_amd as A
from gpu_helpers import tap_errors, golden_inputs

out = {}
cases = [("tiny", A.config_tiny()), ("c1", A.config_c1())]
for tag, cfg in cases:
    x, t = golden_inputs(tag)
    for dtype in ("fp32", "bf16"):
        for flags in (1, 0):
            key = f"{tag}/{dtype}/{'sep' if flags else 'fused'}"
            t0 = time.time()
            try:
                errs, y, yo = tap_errors(cfg, x, t, dtype, flags)
                out[key] = errs
                worst = max(errs, key=errs.get)
                print(f"{key}: out={errs['out']:.3e} worst={worst}:{errs[worst]:.3e} ({time.time()-t0:.1f}s)", flush=True)
                first_bad = next((k for k, v in errs.items() if not (v < (1e-3 if dtype == 'fp32' else 1e-1))), None)
                if first_bad:
                    print("   first tap over tolerance:", first_bad, errs[first_bad], flush=True)
                    print("   " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)
            except Exception as e:
                traceback.print_exc()
                out[key] = {"error": str(e)}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/diag_taps.json", "w"), indent=1)
