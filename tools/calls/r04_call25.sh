#!/bin/bash
# round 4, call 25: socket power and reported sclk (rocm-smi, every ~0.25 s) while configs[4] runs, without and with the next-tile L2 prefetch of the layer kernel
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c25; mkdir -p $out
for v in 0 1; do
  ADF_WN_PREFETCH=$v bash tools/power_probe.sh $out/power_pf$v.raw -- timeout -k 10 400 python3 bench.py --config c5 --steps 6 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_pf$v.json 2> $out/bench_pf$v.err || { tail -3 $out/bench_pf$v.err; exit 1; }
  python3 - $v $out <<'PY'
import re, sys, json, statistics
v, out = sys.argv[1], sys.argv[2]
rows = []
for l in open(f"{out}/power_pf{v}.raw"):
    p = re.search(r"Power \(W\): ([\d.]+)", l); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if p and c: rows.append((float(p.group(1)), int(c.group(1))))
busy = [r for r in rows if r[0] > 700]
d = json.loads(open(f"{out}/bench_pf{v}.json").read().strip().splitlines()[-1])
print(f"ADF_WN_PREFETCH={v}: {d['ms_per_step']:.1f} ms per step; {len(busy)} samples under load: power median {statistics.median(r[0] for r in busy):.0f} W (max {max(r[0] for r in busy):.0f}), reported sclk median {statistics.median(r[1] for r in busy)} MHz")
PY
done
