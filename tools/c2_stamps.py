"""Phase timeline of one workgroup of the spatial-tile conv2d kernel (diagnostic build with -DADF_C2_STAMP).
usage: tools/build_variant.sh c2stamp -DADF_C2_STAMP ; ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_c2stamp.so python tools/c2_stamps.py
Runs one eager pass of the config-4 net at batch 16: the stamps that remain are those of the LAST tile-kernel launch with a stamped block
(the final ResBlock's second conv at the first level: 128 -> 128 channels, 80 x 256)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiodiffuser_amd as A
from audiodiffuser_amd.adm_config import generate_weights
dev = torch.device("cuda", 0)
cfg = A.config_c4()
net = A.UNetModel.from_config(cfg, compute_dtype="bf16")
net.load_state_dict(generate_weights(cfg, seed=0))
net = net.to(dev)
x = torch.randn(16, 1, 80, 256, device=dev)
net(x, torch.zeros(16, device=dev)); net(x, torch.zeros(16, device=dev))
torch.cuda.synchronize()
hd = net.native(dev)
buf = (C.c_ulonglong * 128)()
fn = hd.lib.adf_debug_c2_stamps
fn.restype = C.c_int
print("copy rc", fn(buf))
names = ["entry", "first loads issued + table barrier", "first stage stored", "chunk 0 done (9 taps)", "(unused)", "chunk 1 done", "loop done", "tile in LDS", "end"]
t0 = min(buf[w * 16] for w in range(8) if buf[w * 16])
print("%-36s" % "point" + "".join("%8s" % ("w%d" % w) for w in range(8)))
for i, nm in enumerate(names):
    print("%-36s" % nm + "".join("%8d" % (buf[w * 16 + i] - t0 if buf[w * 16 + i] else -1) for w in range(8)))
