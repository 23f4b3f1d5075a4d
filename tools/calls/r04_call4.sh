#!/bin/bash
# round 4, call 4: the floor microbenchmark of the rb sub-step; split-bf16 attention kernel: tests + step time + per-kernel table
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c4; mkdir -p $out
echo "== rb_floor"; timeout -k 10 120 tools/micro/bin/rb_floor | tee $out/rb_floor.txt
timeout -k 10 120 tools/micro/bin/rb_floor | tail -7 | tee -a $out/rb_floor.txt
echo "== f32x3 tests"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "f32x3 or config2_full_sampler or module_call_site" 2>&1 | tail -8
echo "== f32x3 bench"; timeout -k 10 400 python bench.py --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads 2>/dev/null | tail -1 | cut -c1-300
echo "== f32x3 kernel trace"; rm -rf /tmp/p1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 198 --grid > $out/f32x3_per_nfe_summary.txt; head -16 $out/f32x3_per_nfe_summary.txt
