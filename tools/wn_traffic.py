"""Per-layer HBM traffic of the WaveNetNoise residual-layer kernel (BASELINE configs[4]) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
which layers (dilation 2^(n mod 12)) move more than their algorithmic bytes.
usage (GPU box): python tools/wn_traffic.py DIR_FETCH DIR_WRITE [batch] [T] [C]"""
import collections, csv, glob, sys
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
T = int(sys.argv[4]) if len(sys.argv) > 4 else 22050
Cc = int(sys.argv[5]) if len(sys.argv) > 5 else 256


def per_dispatch(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "wn_layer" in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), 0.0)
            per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    return [v for _, v in sorted(per.items())]


fe, wr = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
n = min(len(fe), len(wr)) // 36 * 36
algo = B * T * Cc * (2 + 2 + 8) + 8 * Cc * Cc * 2          # y read, y_next write (bf16), fp32 skip read-modify-write; + the layer's weights once
print(f"# {len(fe)} / {len(wr)} wn_layer dispatches; per layer: mean over {n // 36} passes; read = 2 * FETCH_SIZE KB (gfx950 half-count of 16-B/lane reads), write = WRITE_SIZE KB")
print(f"# algorithmic bytes per launch: {algo / 1e9:.3f} GB (B = {B}, T = {T}, C = {Cc})")
print("layer dilation   read GB  write GB  total GB   x algorithmic")
tot = 0.0
for l in range(36):
    rf = sum(fe[p * 36 + l] for p in range(n // 36)) / (n // 36) * 2 * 1024 / 1e9
    wf = sum(wr[p * 36 + l] for p in range(n // 36)) / (n // 36) * 1024 / 1e9
    tot += rf + wf
    print(f"{l:5d} {2 ** (l % 12):8d} {rf:9.3f} {wf:9.3f} {rf + wf:9.3f} {(rf + wf) * 1e9 / algo:10.3f}")
print(f"mean per layer {tot / 36:.3f} GB = {tot / 36 * 1e9 / algo:.3f} x algorithmic")
