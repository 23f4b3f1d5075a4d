"""GPU diagnostic: the fp32 ADM UNetModel path against the oracle at B = 8, 32 x 64 (prints one JSON object).  Run as a child process by
tests/test_adm.py with ADF_CONV2D_TILE=0 / 1: the route switch of the same-size 3x3 convs (spatial-tile kernel vs per-tap gather kernel) is read
once per process."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import audiodiffuser_amd as A
from audiodiffuser_amd.adm_config import generate_weights
from oracle import unet2d_oai as O

cfg = A.config_c4_small()
w = generate_weights(cfg, seed=11)
g = torch.Generator().manual_seed(6)
x, t = torch.randn(8, 1, 32, 64, generator=g), torch.linspace(-1.0, 1.0, 8)
net = A.UNetModel.from_config(cfg, compute_dtype="fp32")
net.load_state_dict(w)
y = net.cuda()(x.cuda(), t.cuda()).cpu()
with torch.no_grad():
    ref = O.unet2d_forward(w, cfg, x, t)
print(json.dumps({"route_tile": os.environ.get("ADF_CONV2D_TILE", "1"),
                  "max_rel": float((y - ref).abs().max() / ref.abs().max())}))
