#!/bin/bash
# usage (GPU box): tools/trace_grep.sh "pattern|pattern"   -> per-NFE kernel times of one bench step matching the pattern
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/tg; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tg -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/tg.log 2>&1 || { tail -3 /tmp/tg.log; exit 1; }
python3 tools/trace_summary.py $(ls /tmp/tg/*/*kernel_trace.csv | head -1) 198 --grid | grep -E -i "per NFE|$1"
