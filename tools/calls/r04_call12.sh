#!/bin/bash
# round 4, call 12: knock-outs of the producer / consumer split-bf16 kernel (timing only): where does a launch go?
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c12; mkdir -p $out
V=audiodiffuser_amd/build/variants
run() {
  rm -rf /tmp/pk
  ADF_HIP_LIB=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 bench.py --dtype f32x3 --steps 1 --warmup 0 --num-steps 4 --no-graph --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/pk.log 2>&1 || { tail -3 /tmp/pk.log; return; }
  python3 - "$1" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pk/*/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'x3p' in r['Kernel_Name']:
        d[r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(sys.argv[1], ' '.join(f"g{g}: n={len(v)} mean {sum(v)/len(v):.1f} us" for g, v in sorted(d.items(), key=lambda kv: -int(kv[0]))))
PY
}
run full audiodiffuser_amd/libadf_hip.so
run no_mfma $V/libadf_hip_x3k1.so
run no_act_staging $V/libadf_hip_x3k2.so
run no_weight_dma $V/libadf_hip_x3k4.so
run no_epilogue_stores $V/libadf_hip_x3k8.so
run none_of_them $V/libadf_hip_x3k15.so
