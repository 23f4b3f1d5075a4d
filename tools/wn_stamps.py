"""Phase timeline of one workgroup of the bf16 WaveNet layer kernel (diagnostic build with -DADF_WN_STAMP).

usage: tools/build_variant.sh wnstamp -DADF_WN_STAMP   (build container; the .so travels)
       ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_wnstamp.so python tools/wn_stamps.py [layer] [batch]
Prints, for the 8 waves of the workgroup in the middle of the grid, s_memtime (cycles from the first wave's entry) at the phase
boundaries of the layer kernel, for the last replayed launch."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_wavenet_weights

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 5
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
T = 22050
TMS = 64 if os.environ.get("ADF_WN_WIDE") == "0" else 128
dev = torch.device("cuda", 0)
cfg = A.config_c5()
net = A.WaveNetNoise.from_config(cfg, compute_dtype="bf16")
net.load_state_dict(generate_wavenet_weights(cfg, seed=0))
net = net.to(dev)
net(torch.randn(batch, T, device=dev), torch.zeros(batch, device=dev))
hd = net.native(dev)
ms, by, fl = C.c_float(), C.c_double(), C.c_double()
rc = hd.lib.adf_bench_wavenet_layer(hd.h, batch, T, layer, 5, C.byref(ms), C.byref(by), C.byref(fl), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
torch.cuda.synchronize()
print("rc", rc, "layer", layer, "dilation", cfg.dilation(layer), "ms", ms.value, "TF/s", fl.value / ms.value / 1e9, "us per tile and CU",
      ms.value * 1e3 * 256 / (batch * ((T + TMS - 1) // TMS)))
buf = (C.c_ulonglong * 128)()
fn = hd.lib.adf_debug_wn_stamps
fn.restype = C.c_int
print("copy rc", fn(buf))
names = ["entry", "staging issued", "barrier 1", "GEMM 1 done", "gate stored", "barrier 2", "GEMM 2 done", "epilogue done (wide: y_next pass done)", "barrier 3", "end"]
t0 = min(buf[w * 16] for w in range(8))
print("%-16s" % "point" + "".join("%8s" % ("w%d" % w) for w in range(8)))
for i, nm in enumerate(names):
    print("%-16s" % nm + "".join("%8d" % (buf[w * 16 + i] - t0) for w in range(8)))
print("total span %d ticks for %.1f us per tile: tick = %.2f ns if one workgroup per CU" % (max(buf[w*16+9] for w in range(8)) - t0, ms.value * 1e3 * 256 / (batch * ((T + TMS - 1) // TMS)), ms.value * 1e6 * 256 / (batch * ((T + TMS - 1) // TMS)) / (max(buf[w*16+9] for w in range(8)) - t0)))
