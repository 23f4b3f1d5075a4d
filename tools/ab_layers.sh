#!/bin/bash
# usage: tools/ab_layers.sh VAR v1 v2 ...  -> per-resblock replay rows (in-pass form, rotating operands) for each value of env var VAR
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v ADF_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --roofline-only --roofline-iters 30 2>/dev/null | tail -1 > /tmp/rr_$v.json
  python3 - /tmp/rr_$v.json <<'PY'
import json,sys
rows=json.load(open(sys.argv[1]))["rows"]
tot=0
for r in rows:
    tot+=r["ms"]
    print(f"rb{r['resblock']:2d} conv{r['kernel']} {r['ms']*1e3:7.1f} us {r['bytes']/r['ms']/1e6:6.0f} GB/s {r['flops']/r['ms']/1e9:6.0f} TF/s")
print(f"total {tot*1e3:.1f} us")
PY
done
