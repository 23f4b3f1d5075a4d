"""Per-wave timeline of one steady-state 3-tap K block of the resblock conv kernel (diagnostic build with -DADF_RB_STAMP).

usage (GPU box): tools/build_variant.sh rbstamp -DADF_RB_STAMP -fno-slp-vectorize   (in the build container, the .so travels)
                 ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbstamp.so python tools/rb_stamps.py [resblock] [conv]
Prints, for the 8 waves of thread block 0, the s_memtime stamps (cycles from the first wave's start of the block) at the phase
boundaries of K block 2 of tile 1."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_weights

rb = int(sys.argv[1]) if len(sys.argv) > 1 else 27
conv = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
cfg = A.PRESETS["c2"]()
net = A.UNet1dBase.from_config(cfg, compute_dtype="bf16")
net.load_state_dict(generate_weights(cfg, seed=0))
net = net.to(dev)
x = torch.randn(64, 1, 16384, device=dev)
net(x, torch.zeros(64, device=dev))
hd = net.native(dev)
lib = hd.lib
ms, by, fl, cp = C.c_float(), C.c_double(), C.c_double(), C.c_int()
stream = torch.cuda.current_stream(dev).cuda_stream
rc = lib.adf_bench_layer(hd.h, 64, 16384, rb, conv, 10, C.byref(ms), C.byref(by), C.byref(fl), C.byref(cp), C.c_void_p(stream))
torch.cuda.synchronize()
print("rc", rc, "resblock", rb, "conv", conv, "us", ms.value * 1e3)
buf = (C.c_ulonglong * 128)()
fn = lib.adf_debug_rb_stamps
fn.restype = C.c_int
print("copy rc", fn(buf))
names = ["block start", "u0 mfma done", "u0 dma wait", "u0 barrier", "u1 mfma done", "u1 dma wait", "u1 barrier", "u2 mfma done", "u2 dma wait",
         "u2 barrier"]
t0 = min(buf[w * 16] for w in range(8))
print("%-18s" % "point" + "".join("%8s" % ("w%d" % w) for w in range(8)))
for i, nm in enumerate(names):
    print("%-18s" % nm + "".join("%8d" % (buf[w * 16 + i] - t0) for w in range(8)))
e0 = min(buf[w * 16 + 11] for w in range(8))
print("kernel timeline (cycles from the first wave's entry)")
for i, nm in ((11, "entry"), (12, "first DMAs issued"), (13, "landed + sync"), (14, "first block ready"), (15, "tile 0 done (epilogue)"), (0, "stamped K block")):
    print("%-24s" % nm + "".join("%8d" % (buf[w * 16 + i] - e0) for w in range(8)))
