#!/bin/bash
# round 4, call 15: per-NFE kernel table of the f32x3 mode with adf_gemm_rbx3.h
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c15; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p8.log 2>&1 && python3 tools/trace_summary.py $(ls /tmp/p8/*/*kernel_trace.csv | head -1) 198 --grid | sed "s#/tmp/p8/[^ ]*#rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1#" > $out/f32x3_mode_per_nfe_summary.txt
head -30 $out/f32x3_mode_per_nfe_summary.txt
