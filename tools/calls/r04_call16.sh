#!/bin/bash
# round 4, call 16: f32x3 mode with the f = 2 transposed convs in their 3-tap form on adf_gemm_rbx3.h and the split-bf16 to_out kernel: parity, then the step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c16; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_bf16 or f32x3" > $out/pytest.log 2>&1; rc=$?
tail -25 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p8.log 2>&1 && python3 tools/trace_summary.py $(ls /tmp/p8/*/*kernel_trace.csv | head -1) 198 --grid | sed "s#/tmp/p8/[^ ]*#rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1#" > $out/f32x3_mode_per_nfe_summary.txt
head -16 $out/f32x3_mode_per_nfe_summary.txt
timeout -k 10 300 python3 bench.py --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_f32x3.json 2> $out/bench_f32x3.err || { tail -5 $out/bench_f32x3.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$out/bench_f32x3.json').read().strip().splitlines()[-1]); print('f32x3', d['ms_per_step'], d.get('roofline',{}).get('frac'))"
