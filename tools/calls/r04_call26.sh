#!/bin/bash
# round 4, call 26: attention at 256 tokens (configs[1]: 2 launches per evaluation) with eight waves per workgroup = one workgroup per (sample, head) pair (ADF_ATT_NW=9) against four (=8, default)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c26; mkdir -p $out
for nw in 9 8 9 8; do
rm -rf /tmp/pt
ADF_ATT_NW=$nw timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 bench.py --steps 1 --warmup 0 --num-steps 8 --no-cpu-baseline --no-graph --no-pmc --no-precision-check --no-other-workloads > /tmp/pt.log 2>&1 || { tail -5 /tmp/pt.log; exit 1; }
python3 - $nw <<'PY'
import csv, glob, sys, statistics
f = glob.glob('/tmp/pt/*/*kernel_trace.csv')[0]
d = {}
for r in csv.DictReader(open(f)):
    if 'attention_mfma32' in r['Kernel_Name']:
        d.setdefault(r['Kernel_Name'][:48] + ' wg' + r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')) + ' g' + r.get('Grid_Size_X', r.get('Grid_Size', '?')), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print('ADF_ATT_NW=' + sys.argv[1], k, 'n=%d median %.1f us' % (len(v), statistics.median(v)))
PY
done
ADF_ATT_NW=9 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention or attn or unfused or c3" 2>&1 | tail -2
