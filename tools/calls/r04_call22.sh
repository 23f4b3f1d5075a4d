#!/bin/bash
# round 4, call 22: WaveNet layer kernel with the next-tile L2 prefetch (ADF_WN_PREFETCH=1, default) against without (=0): WaveNet GPU tests, A/B of configs[4],
# phase timelines of both (stamped variant build)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c22; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_wavenet.py -x -q -m gpu > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do
  ADF_WN_PREFETCH=$v timeout -k 10 400 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_c5_pf$v.json 2> $out/bench_c5_pf$v.err || { tail -5 $out/bench_c5_pf$v.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/bench_c5_pf$v.json').read().strip().splitlines()[-1]); print('ADF_WN_PREFETCH=$v', round(d['ms_per_step'],1), [(r['layer'], round(r['ms'],3)) for r in d['roofline']['rows']])"
done
for v in 1 0; do
  for layer in 0 5 11; do
    ADF_WN_PREFETCH=$v ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_wnstamp.so timeout -k 10 200 python3 tools/wn_stamps.py $layer 128 >> $out/wn_stamps_pf$v.txt 2>&1 || { tail -5 $out/wn_stamps_pf$v.txt; exit 1; }
  done
done
head -14 $out/wn_stamps_pf1.txt
