#!/bin/bash
# usage: tools/ab_bench.sh VAR v1 v2 ...   -> ms_per_step of bench.py for each value of env var VAR (same box, same call)
var=$1; shift
for v in "$@"; do
  ms=$(env $var=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],1))')
  echo "$var=$v ms_per_step=$ms"
done
