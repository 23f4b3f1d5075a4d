"""Whole-launch timeline of the resblock conv kernel (diagnostic build with -DADF_RB_TL): where the cycles of ONE launch go.

usage (GPU box): tools/build_variant.sh rbtl -DADF_RB_TL -fno-slp-vectorize        (in the build container; the .so travels)
                 ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbtl.so python tools/rb_timeline.py [resblock] [conv]
For thread block 0 and one block from the middle of the grid: per wave, s_memtime at entry / first DMAs issued / landed + barrier /
first K block ready, per tile the epilogue of the previous tile and every sub-step ("MFMAs and gap work done", "barrier passed"),
the last epilogue; s_memrealtime at both ends gives the clock.  Printed: the per-phase table (cycles, slowest wave) and the
share of the launch inside / outside the K loop."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiodiffuser_amd as A
from audiodiffuser_amd.weights import generate_weights

rb = int(sys.argv[1]) if len(sys.argv) > 1 else 28
conv = int(sys.argv[2]) if len(sys.argv) > 2 else 1
force_single = len(sys.argv) > 4 and sys.argv[4] == "single"      # one tile per thread block with more than 12 sub-steps (the L = 256 launches of adf_gemm_pp.h)
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
rc = lib.adf_bench_layer(hd.h, 64, 16384, rb, conv, 20, C.byref(ms), C.byref(by), C.byref(fl), C.byref(cp), C.c_void_p(stream))
torch.cuda.synchronize()
print(f"rc {rc} resblock {rb} conv {conv}: {ms.value * 1e3:.1f} us per launch in this (stamped) build, {by.value / 1e6:.1f} MB algorithmic")
buf = (C.c_uint * (2 * 8 * 128))()
fn = lib.adf_debug_rb_timeline
fn.restype = C.c_int
assert fn(buf) == 0
M = 1 << 32


def d(a, b):
    return (a - b) % M


for which in (0, 1):
    S = [[buf[(which * 8 + w) * 128 + i] for i in range(128)] for w in range(8)]
    if not any(S[0]):
        continue
    t0 = min(S[w][0] for w in range(8))         # (stamps are within 2^31 of each other: the minimum is the earliest)
    end = max(d(S[w][109], t0) for w in range(8))
    real0 = S[0][120] | (S[0][121] << 32)
    real1 = max(S[w][122] | (S[w][123] << 32) for w in range(8))
    us = (real1 - real0) / 100.0
    print(f"\n=== thread block {'0' if which == 0 else 'grid/2 + 3'}: {end} cycles entry -> last epilogue done, {us:.1f} us by s_memrealtime => {end / us / 1e3:.2f} GHz")
    rows = []

    def phase(name, a, b):
        """cycles from stamp a to stamp b, per wave; the figure kept is the slowest wave's"""
        v = [d(S[w][b], S[w][a]) for w in range(8)]
        rows.append((name, max(v), min(v)))
        return max(v)

    out_k = 0
    out_k += phase("entry -> first DMAs issued (bias / table loads, descriptors)", 0, 1)
    out_k += phase("first DMAs issued -> landed + barrier", 1, 2)
    out_k += phase("landed -> first K block activated + barrier", 2, 3)
    in_k = 0
    ntile = 0
    for t in range(1 if force_single else 4):
        base = 4 + t * 26
        if not S[0][base + 3] or (t > 0 and not S[0][base]):
            break
        ntile += 1
        if t > 0:
            out_k += phase(f"tile {t}: epilogue of tile {t - 1} (accumulators -> LDS -> 16-byte stores, statistics)", base, base + 1)
        prev = base + 1 if t > 0 else 3
        nsub = 0
        single = force_single or (t == 0 and not S[0][4 + 26 + 3])           # one tile per block (n = 256 launches at L = 1024): its sub-steps may run past 12
        while nsub < (50 if single else 12) and base + 3 + 2 * nsub < 108 and S[0][base + 3 + 2 * nsub]:
            nsub += 1
        work = [d(S[w][base + 2 + 2 * (nsub - 1) + 1], S[w][prev]) for w in range(8)]
        in_k += max(work)
        waits = [sum(d(S[w][base + 3 + 2 * u], S[w][base + 2 + 2 * u]) for u in range(nsub)) for w in range(8)]
        rows.append((f"tile {t}: {nsub} sub-steps (K loop)", max(work), min(work)))
        rows.append((f"   of which: wait for DMA + barrier at the end of the sub-steps", max(waits), min(waits)))
        per = [max(d(S[w][base + 3 + 2 * u], S[w][base + 3 + 2 * (u - 1)] if u else S[w][prev]) for w in range(8)) for u in range(nsub)]
        rows.append(("   per sub-step: " + " ".join(str(p) for p in per), 0, 0))
    out_k += phase("last epilogue", 108, 109)
    for name, mx, mn in rows:
        print(f"{name:100s} {mx:8d} {mn:8d}" if mx or mn else name)
    print(f"inside the K loops: {in_k} cycles = {100.0 * in_k / end:.1f} % of the launch; outside (start-up, epilogues): {out_k} = {100.0 * out_k / end:.1f} %")
    # per wave: entry skew, and for tile 1 the work (previous barrier -> MFMAs + gap work done) and the wait (-> barrier passed) of every sub-step
    print("per wave (w0 .. w7)")
    print(f"{'entry after the first wave':40s}" + "".join(f"{d(S[w][0], t0):7d}" for w in range(8)))
    for nm, a, b in (("entry -> DMAs issued + table stored", 0, 1), ("-> landed + barrier", 1, 2), ("-> first block activated", 2, 3)):
        print(f"{nm:40s}" + "".join(f"{d(S[w][b], S[w][a]):7d}" for w in range(8)))
    base = 4 if force_single else 4 + 26
    if not force_single:
        print(f"{'tile 1: epilogue of tile 0':40s}" + "".join(f"{d(S[w][base + 1], S[w][base]):7d}" for w in range(8)))
    for u in range(nsub if force_single else 12):
        prev = (3 if force_single else base + 1) if u == 0 else base + 3 + 2 * (u - 1)
        print(f"{'tile %d sub-step %2d work' % (0 if force_single else 1, u):40s}" + "".join(f"{d(S[w][base + 2 + 2 * u], S[w][prev]):7d}" for w in range(8)))
        print(f"{'                   wait + barrier':40s}" + "".join(f"{d(S[w][base + 3 + 2 * u], S[w][base + 2 + 2 * u]):7d}" for w in range(8)))
    if any(S[w][113] for w in range(8)) and not any(S[w][114] for w in range(8)):
        print("start-up, fine (cycles from the wave's entry): argument head used / statistics + parameter loads issued / first DMAs issued / loaded values there / table stored")
        for w in range(8):
            print(f"   w{w}: " + " / ".join(str(d(S[w][i], S[w][0])) for i in (110, 111, 112, 113, 1)))
    elif any(S[w][110] for w in range(8)):
        print("fine stamps (role-split kernel): consumers = last epilogue: entry / stores issued / statistics done; producers = tile 1 block 1:")
        print("   per sub-step u: DMAs issued / prologue done / slab wait done      (cycles from the sub-step's start = previous barrier)")
        for w in range(8):
            if not S[w][113]:
                print(f"   w{w}: stores {d(S[w][111], S[w][110])}  statistics {d(S[w][112], S[w][111])}  whole epilogue + bias {d(S[w][109], S[w][108])}")
            else:
                base = 4 + 26
                row = []
                for u in range(3):
                    k = 3 + u           # sub-steps 3..5 of the tile = block 1
                    prev = base + 3 + 2 * (k - 1)
                    row.append("u%d: %d / %d / %d (end %d)" % (u, d(S[w][110 + 3 * u], S[w][prev]), d(S[w][111 + 3 * u], S[w][prev]), d(S[w][112 + 3 * u], S[w][prev]), d(S[w][base + 2 + 2 * k], S[w][prev])))
                print(f"   w{w}: " + "   ".join(row))
import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"resblock": rb, "conv": conv, "us_per_launch_stamped_build": ms.value * 1e3, "stamps": list(buf)}, open(f"gpurun_out/rb_timeline_raw_{rb}_{conv}.json", "w"))
