#!/bin/bash
# round 4, call 8: config 4: 160-pixel tiles where 128-pixel ones divide (A/B, experiments build); config 5: HBM traffic per residual layer (PMC)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c8; mkdir -p $out
V=audiodiffuser_amd/build/variants
echo "== c4 A/B"
for p5 in 0 128 256 1024 0; do
  ADF_HIP_LIB=$V/libadf_hip_exp.so ADF_CONV2D_PREFER5=$p5 timeout -k 10 400 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-precision-check 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('PREFER5=$p5', round(d['ms_per_step'],1), r.get('pass_ms'), round(r['frac'],3))"
done | tee $out/c4_ab.txt
echo "== c5 per-layer traffic"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/wn_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --kernel-include-regex wn_layer --pmc $c --output-format csv -d /tmp/wn_$c -- python3 bench.py --config c5 --steps 1 --warmup 0 --num-steps 2 --no-graph --no-cpu-baseline --no-pmc --no-precision-check > /tmp/wn_$c.log 2>&1 || { tail -5 /tmp/wn_$c.log; exit 1; }
done
python3 tools/wn_traffic.py /tmp/wn_FETCH_SIZE /tmp/wn_WRITE_SIZE | tee $out/c5_traffic_per_layer.txt
