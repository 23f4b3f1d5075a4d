#!/bin/bash
# round 4, call 14: adf_gemm_rbx3.h at small batches against the fp32 oracle (every shape incl. the 128-row form), the f32x3 tests and C2 sampler,
# then the default bench without the child passes (its f32x3_mode leg compares with exact fp32 at batch 64)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c14; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_bf16 or f32x3" > $out/pytest.log 2>&1; rc=$?
tail -25 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-pmc --no-other-workloads > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 -c "import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('f32x3_mode'), d.get('bf16_vs_fp32'))"
