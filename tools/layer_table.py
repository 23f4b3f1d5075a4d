"""Per-launch table of one network pass: the launcher's route lines (ADF_GEMM_TRACE=1, stderr) matched in order with the
conv_gemm rows of a rocprofv3 kernel trace.
usage (GPU box):  ADF_GEMM_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 bench.py --steps 1 --warmup 0 \
                      --num-steps 2 --no-graph --no-cpu-baseline 2> /tmp/lt.err ;  python3 tools/layer_table.py /tmp/lt /tmp/lt.err"""
import csv, glob, re, sys
trace = glob.glob(f"{sys.argv[1]}/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
gem = [r for r in rows if "conv_gemm" in r["Kernel_Name"]]
routes = [l for l in open(sys.argv[2]) if l.startswith("[adf gemm]")]
# the sampler's network passes come first (identical route sequences); bench.py's resblock replay follows: find the period
per = next(p for p in range(20, len(routes)) if routes[:p] == routes[p:2 * p])
print(f"# {len(gem)} conv_gemm launches, {len(routes)} route lines, {per} per network pass; second pass shown")
gem, routes = gem[per:2 * per], routes[per:2 * per]
tot = 0.0
for r, l in zip(gem, routes):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    m = re.search(r"\] (\w+)\s+B=(\d+) lin=(\d+) mrows=(\d+) n=(\d+)/\d+ nseg=(\d+) seg0\(c=(\d+)\+(\d+) taps=(\d+) stride=(\d+)", l)
    route, B, lin, mrows, n, nseg, c0, c1, taps, stride = m.groups()
    s1 = re.search(r"seg1\(c=(\d+)\+(\d+) taps=(\d+)", l)
    k = (int(c0) + int(c1)) * int(taps) + ((int(s1.group(1)) + int(s1.group(2))) * int(s1.group(3)) if s1 else 0)
    if "res=1" in l and route == "pp": k += int(n)
    fl = 2.0 * int(B) * int(mrows) * int(n) * k
    sc = re.search(r"scatter=(\d+)", l).group(1)
    print(f"{route:6s} L={lin:>5s} rows={mrows:>5s} K={k:5d} N={n:>4s} taps={taps} scatter={sc} {us:7.1f}us {fl / us / 1e6:7.1f} TF/s")
print(f"total {tot / 1e3:.3f} ms")
