#!/bin/bash
# round 4, call 3: start-up fill in two parts + slab ahead of the epilogue: parity first, then A/B against the old order and the wave-priority variants, timelines
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c3; mkdir -p $out
echo "== parity (rb kernel routes)"
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config2_batch64_full_length or resblock_dma_kernel or batch64_bf16_is_batch_independent or config2_batch8 or transposed_conv or group_size or module_call_site" 2>&1 | tail -6 || exit 1
V=audiodiffuser_amd/build/variants
ab() {  # name lib
  ms=$(ADF_HIP_LIB=$2 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["ms_per_step"],2), r["ms_per_launch"], round(r["frac"],3), r.get("all_resblocks",{}).get("frac"))')
  echo "$1: $ms"
}
for rep in 1 2; do
  ab product audiodiffuser_amd/libadf_hip.so
  ab old $V/libadf_hip_old.so
  ab fill2only $V/libadf_hip_fill2only.so
  ab prio1 $V/libadf_hip_rbprio1.so
  ab prio2 $V/libadf_hip_rbprio2.so
  ab prio3 $V/libadf_hip_rbprio3.so
done | tee $out/ab.txt
for spec in "28 1" "0 1" "28 2"; do
  set -- $spec
  ADF_HIP_LIB=$V/libadf_hip_rbtl.so timeout -k 10 200 python tools/rb_timeline.py $1 $2 2>&1 | grep -v amdgpu.ids > $out/rb_timeline_$1_$2.txt || exit 1
  head -26 $out/rb_timeline_$1_$2.txt | cut -c1-150
done
