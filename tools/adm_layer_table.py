"""Per-launch table of one ADM U-Net pass: the launcher's route lines (ADF_C2_TRACE=1, stderr) matched in order with the conv2d rows
(conv2d_tile_kernel / conv2d_gemm_kernel) of a rocprofv3 kernel trace.
usage (GPU box):  ADF_C2_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/adm_pass.py 64 2> /tmp/lt.err ;
                  python3 tools/adm_layer_table.py /tmp/lt /tmp/lt.err"""
import csv, glob, re, sys
trace = glob.glob(f"{sys.argv[1]}/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
conv = [r for r in rows if "conv2d_tile_kernel" in r["Kernel_Name"] or "conv2d_gemm_kernel" in r["Kernel_Name"]]
routes = [l for l in open(sys.argv[2]) if l.startswith("[adf conv2d]")]
per = next(p for p in range(10, len(routes)) if routes[:p] == routes[p:2 * p])
print(f"# {len(conv)} conv2d launches, {len(routes)} route lines, {per} per network pass; last pass shown")
conv, routes = conv[-per:], routes[-per:]
other = {}
t_first, t_last = int(conv[0]["Start_Timestamp"]), int(conv[-1]["End_Timestamp"])
for r in rows:
    if r in conv or not (t_first <= int(r["Start_Timestamp"]) <= t_last):
        continue
    k = r["Kernel_Name"].replace("adf::", "").replace("void ", "").split("(")[0].split("<")[0]
    other.setdefault(k, [0, 0.0])
    other[k][0] += 1; other[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0.0
cls = {}
for r, l in zip(conv, routes):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    m = re.search(r"\] (\w+)\s+B=(\d+) H=(\d+) W=(\d+) cin=(\d+) c0=(\d+) cout=(\d+) taps=(\d+) mode=(\d+) ab=(\d) act=(\d) res=(\d) stats=(\d)", l)
    route, B, H, W, cin, c0, cout, taps, mode, ab, act, res, st = m.groups()
    fl = 2.0 * int(B) * int(H) * int(W) * int(cin) * int(cout) * int(taps)
    el = 2
    by = int(B) * int(H) * int(W) * (int(cin) / (4 if mode == "1" else 1) * (4 if mode == "2" else 1) + int(cout) * (2 if res == "1" else 1)) * el + int(cin) * int(cout) * int(taps) * el
    print(f"{route:5s} {H:>3s}x{W:<3s} cin={cin:>4s}{'*' if c0 != cin else ' '} cout={cout:>4s} taps={taps} mode={mode} res={res} {us:8.1f}us {fl / us / 1e6:7.1f} TF/s {by / us / 1e3:7.1f} GB/s")
    key = (route, H, W, cin, cout, taps, mode)
    cls.setdefault(key, [0, 0.0, 0.0]); cls[key][0] += 1; cls[key][1] += us; cls[key][2] += fl
print(f"conv2d total {tot / 1e3:.3f} ms; pass window {(t_last - t_first) / 1e6:.3f} ms")
print("# by class (time-sorted)")
for k, v in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:5s} {k[1]:>3s}x{k[2]:<3s} cin={k[3]:>4s} cout={k[4]:>4s} taps={k[5]} mode={k[6]} n={v[0]:2d} {v[1] / 1e3:7.3f} ms {v[1] / tot * 100:5.1f}% {v[2] / v[1] / 1e6:7.1f} TF/s")
print("# other kernels inside the pass window")
for k, v in sorted(other.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} n={v[0]:3d} {v[1] / 1e3:7.3f} ms")
