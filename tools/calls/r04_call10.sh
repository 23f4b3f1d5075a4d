#!/bin/bash
# round 4, call 10: whole GPU suite, smoke, default bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c10; mkdir -p $out
echo "== full GPU suite"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 || exit 1
echo "== default bench"; timeout -k 10 900 python bench.py 2>/dev/null | tail -1 > $out/r04_bench.json; python3 -c "
import json; d=json.loads(open('$out/r04_bench.json').read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['ms_per_launch_batches'], r['frac'], r['traffic'], d.get('fp32_mode_ms_per_step'), d.get('f32x3_mode'))
print({k:(round(v['ms_per_step'],1), round(v['roofline']['frac'],3), v['roofline'].get('traffic')) for k,v in d['other_workloads'].items()})
print(d['cpu_baseline'].get('value'), d.get('gpu_over_cpu'))"
