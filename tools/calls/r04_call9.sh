#!/bin/bash
# round 4, call 9: attention at 1024 tokens: MFMA utilisation against the wall time of an un-instrumented run; then the whole GPU suite
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c9; mkdir -p $out
rm -rf /tmp/pa /tmp/pt
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex attention_mfma32 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d /tmp/pa -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-pmc --no-precision-check > /tmp/pa.log 2>&1 || { tail -5 /tmp/pa.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-pmc --no-precision-check > /tmp/pt.log 2>&1 || { tail -5 /tmp/pt.log; exit 1; }
python3 tools/attention_util.py /tmp/pa /tmp/pt | tee $out/attention_mfma_utilisation.txt
echo "== full GPU suite"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
