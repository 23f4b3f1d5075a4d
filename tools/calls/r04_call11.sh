#!/bin/bash
# round 4, call 11: the producer / consumer split-bf16 GEMM kernel: f32x3 tests, A/B against the generic kernel, per-kernel table
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c11; mkdir -p $out
echo "== f32x3 tests"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32x3 or config2_full_sampler" 2>&1 | tail -8 || exit 1
for v in 1 0 1 0; do
  ADF_GEMM_X3P=$v timeout -k 10 400 python bench.py --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ADF_GEMM_X3P=$v', round(d['ms_per_step'],1))"
done | tee $out/ab_x3p.txt
echo "== f32x3 kernel trace"; rm -rf /tmp/p1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 198 --grid > $out/f32x3_per_nfe_summary.txt; head -20 $out/f32x3_per_nfe_summary.txt
