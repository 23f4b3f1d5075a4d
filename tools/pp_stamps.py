"""Per-wave timeline of one steady-state K block of the DMA kernel (diagnostic build with -DADF_PP_STAMP).

usage (GPU box): tools/build_variant.sh stamp -DADF_PP_STAMP   (in the build container, the .so travels)
                 ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_stamp.so python tools/pp_stamps.py [resblock]
Prints, for the 8 waves of thread block 0, the s_memtime stamps (cycles from the block's first stamp) at the phase
boundaries of K block nb+1 of the resblock's conv1 launch."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_weights

rb = int(sys.argv[1]) if len(sys.argv) > 1 else 27
dev = torch.device("cuda", 0)
cfg = A.PRESETS["c2"]()
net = A.UNet1dBase.from_config(cfg, compute_dtype="bf16")
net.load_state_dict(generate_weights(cfg, seed=0))
net = net.to(dev)
x = torch.randn(64, 1, 16384, device=dev)
net(x, torch.zeros(64, device=dev))
hd = net.native(dev)
lib = hd.lib
ms1, ms2 = C.c_float(), C.c_float()
d = [C.c_double() for _ in range(4)]
stream = torch.cuda.current_stream(dev).cuda_stream
rc = lib.adf_bench_resblock(hd.h, 64, 16384, rb, 20, C.byref(ms1), C.byref(ms2), *[C.byref(v) for v in d], C.c_void_p(stream))
torch.cuda.synchronize()
print("rc", rc, "conv1 us", ms1.value * 1e3, "conv2 us", ms2.value * 1e3)
buf = (C.c_ulonglong * 256)()
fn = lib.adf_debug_pp_stamps
fn.restype = C.c_int
print("copy rc", fn(buf))
names = {0: "start", 1: "S0 early-tr", 2: "S0 mfma", 3: "S0 late-tr", 4: "S0 dma", 5: "S0 bar", 6: "S1 early-tr", 7: "S1 mfma",
         8: "S1 late-tr", 9: "S1 dma", 10: "S1 bar", 11: "S2 early-tr", 12: "S2 mfma", 13: "S2 late-tr", 14: "S2 dma", 15: "S2 bar", 16: "desc"}
pro = {17: "entry", 18: "dma issued", 19: "bias/tab ld", 20: "dma landed", 21: "syncthreads", 22: "transform", 23: "barrier"}
p0 = min(buf[w * 32 + 17] for w in range(8))
print("kernel prologue (cycles from the first wave's entry)")
for i in sorted(pro):
    print("%-12s" % pro[i] + "".join("%8d" % (buf[w * 32 + i] - p0) for w in range(8)))
print("K block nb+1 starts at", min(buf[w * 32] for w in range(8)) - p0)
t0 = min(buf[w * 32] for w in range(8))
print("%-12s" % "point" + "".join("%8s" % ("w%d" % w) for w in range(8)))
for i in sorted(names):
    print("%-12s" % names[i] + "".join("%8d" % (buf[w * 32 + i] - t0) for w in range(8)))
