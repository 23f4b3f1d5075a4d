#!/bin/bash
# round 4, call 13: split-bf16 form of the resblock conv kernel (adf_gemm_rbx3.h): parity of the f32x3 mode with the route on, then A/B of the step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c13; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32x3" > $out/pytest.log 2>&1; rc=$?
tail -15 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  ADF_GEMM_RBX3=$v timeout -k 10 300 python3 bench.py --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_rbx3_$v.json 2> $out/bench_rbx3_$v.err || { tail -5 $out/bench_rbx3_$v.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/bench_rbx3_$v.json').read().strip().splitlines()[-1]); print('ADF_GEMM_RBX3=$v', d['ms_per_step'], d.get('roofline',{}).get('frac'))"
done
