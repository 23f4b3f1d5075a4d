#!/bin/bash
# round 4, call 27: socket power / reported sclk while configs[3] (ADM 2-D U-Net), configs[2] (c3) and the split-bf16 mode of configs[1] run
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c27; mkdir -p $out
probe() {
  tag=$1; shift
  bash tools/power_probe.sh $out/power_$tag.raw -- timeout -k 10 500 python3 bench.py "$@" --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_$tag.json 2> $out/bench_$tag.err || { tail -3 $out/bench_$tag.err; exit 1; }
  python3 - $tag $out <<'PY'
import re, sys, json, statistics
tag, out = sys.argv[1], sys.argv[2]
rows = []
for l in open(f"{out}/power_{tag}.raw"):
    p = re.search(r"Power \(W\): ([\d.]+)", l); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if p and c: rows.append((float(p.group(1)), int(c.group(1))))
busy = [r for r in rows if r[0] > 600]
d = json.loads(open(f"{out}/bench_{tag}.json").read().strip().splitlines()[-1])
print(f"{tag}: {d['ms_per_step']:.1f} ms per step; {len(busy)} samples under load: power median {statistics.median(r[0] for r in busy):.0f} W (max {max(r[0] for r in busy):.0f}), reported sclk median {statistics.median(r[1] for r in busy)} MHz")
PY
}
probe c4 --config c4 --steps 5 --warmup 1
probe c3 --config c3 --sampler dpm --steps 40 --warmup 2
probe c2_f32x3 --dtype f32x3 --steps 10 --warmup 1
probe c2_bf16 --steps 25 --warmup 2
