#!/bin/bash
# round 4, call 2: whole-launch timelines with the fine start-up stamps; the split-bf16 mode: tests, step time, per-kernel table
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c2; mkdir -p $out
for spec in "28 1" "0 1" "0 2" "4 1"; do
  set -- $spec
  ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbtl.so timeout -k 10 200 python tools/rb_timeline.py $1 $2 2>&1 | grep -v amdgpu.ids > $out/rb_timeline_$1_$2.txt || exit 1
  head -30 $out/rb_timeline_$1_$2.txt | cut -c1-150
  grep -A9 "start-up, fine" $out/rb_timeline_$1_$2.txt | head -10
done
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "f32x3 or config2_full_sampler or module_call_site or group_size" 2>&1 | tail -15
echo "== f32x3 bench"; timeout -k 10 400 python bench.py --dtype f32x3 --steps 2 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads 2>/dev/null | tail -1 | cut -c1-400
echo "== f32x3 kernel trace"; rm -rf /tmp/p1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 198 --grid > $out/f32x3_per_nfe_summary.txt; head -24 $out/f32x3_per_nfe_summary.txt
