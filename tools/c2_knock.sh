#!/bin/bash
# Knock-out table of the spatial-tile conv2d kernel: one traced ADM pass (batch 64) per diagnostic build of tools/build_variant.sh kN -DADF_C2_KNOCK=N
# (run from the repo root on the GPU box; prints the per-class lines of tools/adm_layer_table.py for the three main shapes).
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  lib=""; [ "$v" != base ] && lib=$GRAFT_REPO_ROOT/audiodiffuser_amd/build/variants/libadf_hip_$v.so
  rm -rf /tmp/kn_$v
  ADF_HIP_LIB=$lib ADF_C2_TRACE=1 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kn_$v -- python3 $GRAFT_REPO_ROOT/tools/adm_pass.py 64 2> /tmp/kn_$v.err || { echo "$v failed"; tail -3 /tmp/kn_$v.err; continue; }
  python3 $GRAFT_REPO_ROOT/tools/adm_layer_table.py /tmp/kn_$v /tmp/kn_$v.err > /tmp/kn_$v.txt
  echo "== $v: $(grep 'conv2d total' /tmp/kn_$v.txt)"
  sed -n '/by class/,/other kernels/p' /tmp/kn_$v.txt | grep -E "t4 +80x256 cin= 128 cout= 128|t4 +40x128 cin= 256 cout= 256|t5 +10x32  cin= 512 cout= 512|t4 +20x64  cin= 256"
done
