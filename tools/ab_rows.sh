#!/bin/bash
# usage: tools/ab_rows.sh "rb list" variant1 variant2 ...   (variant "-" = product build)
# per-resblock replay timings of library variants (tools/build_variant.sh) inside one gpurun call
sel=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  echo "== $v"
  lib=""; [ "$v" != "-" ] && lib=$root/audiodiffuser_amd/build/variants/libadf_hip_$v.so
  ADF_HIP_LIB=$lib ADF_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --roofline-only --roofline-iters 50 2>/dev/null | tail -1 > /tmp/rr.json && python3 tools/roof_rows.py /tmp/rr.json $sel
done
