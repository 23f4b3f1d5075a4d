#!/bin/bash
# usage: tools/ab_env_rows.sh "rb list" VAR v1 v2 ...  -> per-resblock replay timings for values of an environment variable
sel=$1; var=$2; shift; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v ADF_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --roofline-only --roofline-iters 50 2>/dev/null | tail -1 > /tmp/rr.json && python3 tools/roof_rows.py /tmp/rr.json $sel
done
