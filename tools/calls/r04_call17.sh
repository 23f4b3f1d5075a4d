#!/bin/bash
# round 4, call 17: WaveNet layer kernel as 64-position tiles on four waves, two workgroups per CU (ADF_WN_WIDE=2): parity of the three routes, then A/B of configs[4]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c17; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_wavenet.py -x -q -m gpu -k "both_layer_kernel_routes" > $out/pytest.log 2>&1; rc=$?
tail -15 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in 1 2 1 2; do
  ADF_WN_WIDE=$v timeout -k 10 400 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_c5_wide$v.json 2> $out/bench_c5_wide$v.err || { tail -5 $out/bench_c5_wide$v.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/bench_c5_wide$v.json').read().strip().splitlines()[-1]); print('ADF_WN_WIDE=$v', d['ms_per_step'], d.get('roofline',{}))"
done
