"""GPU diagnostic: ADM UNetModel on 10-row images (the height of the 80-row mel block three levels down), B = 32 at 10 x 128, so that the
same-size 3x3 convs of the first level take the 5 x 32-pixel spatial tiles (160-pixel workgroups, conv2d_tile_kernel<T, 5, 1>).  fp32 against the
oracle; bf16 teacher-forced against the bf16-storage oracle, every stored tensor.  Prints one JSON object; run by tests/test_adm.py as a child
process with ADF_C2_TRACE=1 so that the route lines on stderr prove which kernel ran."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import audiodiffuser_amd as A
from audiodiffuser_amd.adm_config import generate_weights
from oracle import unet2d_oai as O

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(8)
x, t = torch.randn(32, 1, 10, 128, generator=g), torch.linspace(-1.0, 1.0, 32)
out = {}
cfg = A.config_c4_small()
w = generate_weights(cfg, seed=13)
net = A.UNetModel.from_config(cfg, compute_dtype="fp32")
net.load_state_dict(w)
y = net.to(dev)(x.to(dev), t.to(dev)).cpu()
with torch.no_grad():
    ref = O.unet2d_forward(w, cfg, x, t)
out["fp32_max_rel"] = float((y - ref).abs().max() / ref.abs().max())

cfg = A.ADMConfig(image_size=32, in_channels=1, model_channels=64, out_channels=1, num_res_blocks=1, attention_resolutions="16",
                  channel_mult=(1, 2), num_heads=2)
w = generate_weights(cfg, seed=14)
net = A.UNetModel.from_config(cfg, compute_dtype="bf16")
net.load_state_dict(w)
net = net.to(dev)
y = net(x.to(dev), t.to(dev)).cpu()
hd = net.native(dev)
taps = {k: hd.tap(k, 32, dev).cpu() for k in hd.tap_names()}
errs = {}
with torch.no_grad():
    y_f = O.unet2d_forward(w, cfg, x, t, storage="bf16", force=taps, errs=errs)
out["bf16_taps"] = len(errs)
out["bf16_missing"] = sorted(set(taps) ^ set(errs))
out["bf16_worst_conv"] = max([v for k, v in errs.items() if not k.endswith(".att")] + [O.rel_l2(y, y_f)])
out["bf16_worst_att"] = max([v for k, v in errs.items() if k.endswith(".att")] + [0.0])
print(json.dumps(out))
