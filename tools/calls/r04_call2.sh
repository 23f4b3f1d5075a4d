#!/bin/bash
# round 4, call 2: whole-launch timelines with the fine start-up stamps (dominant launch, a K = 384 conv1 and its identity-residual conv2)
set -o pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c2; mkdir -p $out
for spec in "28 1" "0 1" "0 2" "4 1"; do
  set -- $spec
  ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbtl.so timeout -k 10 200 python tools/rb_timeline.py $1 $2 2>&1 | grep -v amdgpu.ids > $out/rb_timeline_$1_$2.txt || exit 1
  head -30 $out/rb_timeline_$1_$2.txt | cut -c1-150
  grep -A9 "start-up, fine" $out/rb_timeline_$1_$2.txt | head -10
done
