#!/bin/bash
# round 4, call 20: attention kernel with eight waves per workgroup at N >= 512 (ADF_ATT_NW=8, default) against four (=4): parity of both, timing of the N = 1024 launches, c3 step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c20; mkdir -p $out
for nw in 8 4; do
ADF_ATT_NW=$nw timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention or attn or unfused or c3" > $out/pytest_nw$nw.log 2>&1; rc=$?
tail -3 $out/pytest_nw$nw.log
[ $rc -eq 0 ] || exit $rc
done
for nw in 8 4 8 4; do
rm -rf /tmp/pt
ADF_ATT_NW=$nw timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-pmc --no-precision-check > /tmp/pt.log 2>&1 || { tail -5 /tmp/pt.log; exit 1; }
python3 - $nw <<'PY'
import csv, glob, sys, statistics
f = glob.glob('/tmp/pt/*/*kernel_trace.csv')[0]
d = {}
for r in csv.DictReader(open(f)):
    if 'attention_mfma32' in r['Kernel_Name']:
        d.setdefault(r['Kernel_Name'][:60] + ' wg' + r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')) + ' g' + r.get('Grid_Size_X', r.get('Grid_Size', '?')), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print('ADF_ATT_NW=' + sys.argv[1], k, 'n=%d median %.1f us' % (len(v), statistics.median(v)))
PY
done
for nw in 8 4; do
ADF_ATT_NW=$nw timeout -k 10 400 python3 bench.py --config c3 --sampler dpm --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_c3_nw$nw.json 2> $out/bench_c3.err || { tail -5 $out/bench_c3.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$out/bench_c3_nw$nw.json').read().strip().splitlines()[-1]); print('c3 ADF_ATT_NW=$nw', d['ms_per_step'])"
done
