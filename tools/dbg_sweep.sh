#!/bin/bash
# usage: tools/dbg_sweep.sh "rb list" dbg1 dbg2 ...  -> per-resblock replay timings under ADF_GEMM_DBG values
# (knock-outs: 1 skip stores, 2 skip prologue math, 4 skip MFMA, 8 skip stats; other ADF_GEMM_* variables pass through)
sel=$1; shift
for v in "$@"; do
  echo "== ADF_GEMM_DBG=$v"
  ADF_GEMM_DBG=$v ADF_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --roofline-only --roofline-iters 50 2>/dev/null | tail -1 > /tmp/rr.json && python3 tools/roof_rows.py /tmp/rr.json $sel
done
