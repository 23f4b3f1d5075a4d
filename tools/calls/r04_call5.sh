#!/bin/bash
# round 4, call 5: progress-based wave priority: microbenchmark and A/B
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c5; mkdir -p $out
echo "== rb_floor"; timeout -k 10 120 tools/micro/bin/rb_floor | tee $out/rb_floor.txt
V=audiodiffuser_amd/build/variants
ab() {  # name lib
  ms=$(ADF_HIP_LIB=$2 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["ms_per_step"],2), r["ms_per_launch_batches"], round(r["frac"],3), round(r["all_resblocks"]["hbm_frac"],3))')
  echo "$1: $ms"
}
for rep in 1 2; do
  ab product audiodiffuser_amd/libadf_hip.so
  ab prio4 $V/libadf_hip_rbprio4.so
done | tee $out/ab.txt
echo "== parity of the prio4 build (same arithmetic: one quick check)"
ADF_HIP_LIB=$V/libadf_hip_rbprio4.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config2_batch64_full_length" 2>&1 | tail -3
