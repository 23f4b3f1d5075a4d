#!/bin/bash
# round 4, call 28: split-bf16 attention at 257 .. 1024 tokens (K from global): parity of the f32x3 tests, then configs[2] in the split-bf16 mode with the kernel on / off (ADF_ATT_X3)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c28; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f32x3 or do_not_fill or split_bf16" > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  ADF_ATT_X3=$v timeout -k 10 500 python3 bench.py --config c3 --sampler dpm --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_c3_x3_$v.json 2> $out/bench_c3_x3_$v.err || { tail -5 $out/bench_c3_x3_$v.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/bench_c3_x3_$v.json').read().strip().splitlines()[-1]); print('c3 f32x3 ADF_ATT_X3=$v', round(d['ms_per_step'],1))"
done
