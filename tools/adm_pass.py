"""Three eager passes of the config-4 ADM U-Net (bf16) at the given batch: the workload behind tools/adm_layer_table.py."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiodiffuser_amd as A
from audiodiffuser_amd.adm_config import generate_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
cfg = A.config_c4()
net = A.UNetModel.from_config(cfg, compute_dtype="bf16")
net.load_state_dict(generate_weights(cfg, seed=0))
net = net.to(dev)
x = torch.randn(B, 1, 80, 256, device=dev)
for _ in range(3):
    net(x, torch.zeros(B, device=dev))
torch.cuda.synchronize()
