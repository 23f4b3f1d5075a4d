#!/bin/bash
# round 4, call 1: the new C2 99-NFE fp32 sampler test; where the fp32 (parity) mode's step goes (kernel stats + per-launch table)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c1; mkdir -p $out
echo "== test"; timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "config2_full_sampler" 2>&1 | tail -5
echo "== fp32 kernel trace"; rm -rf /tmp/p1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --dtype fp32 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
tail -1 /tmp/p1.log | cut -c1-600
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 198 --grid > $out/fp32_per_nfe_summary.txt; head -40 $out/fp32_per_nfe_summary.txt
echo "== fp32 layer table"; rm -rf /tmp/lt
ADF_GEMM_TRACE=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 bench.py --dtype fp32 --steps 1 --warmup 0 --num-steps 2 --no-graph --no-cpu-baseline --no-pmc --no-precision-check --no-other-workloads > /tmp/lt.log 2> /tmp/lt.err || { tail -5 /tmp/lt.err; exit 1; }
python3 tools/layer_table.py /tmp/lt /tmp/lt.err > $out/fp32_layer_table.txt; tail -5 $out/fp32_layer_table.txt
