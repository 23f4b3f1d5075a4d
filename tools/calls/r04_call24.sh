#!/bin/bash
# round 4, call 24: per-NFE kernel table of configs[2] (c3, DPM multistep, 49 evaluations)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c24; mkdir -p $out
rm -rf /tmp/p9
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p9 -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p9.log 2>&1 || { tail -5 /tmp/p9.log; exit 1; }
python3 tools/trace_summary.py $(ls /tmp/p9/*/*kernel_trace.csv | head -1) 98 --grid | sed "s#/tmp/p9/[^ ]*#rocprofv3 --kernel-trace --stats -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 1#" > $out/c3_per_nfe_summary.txt
head -45 $out/c3_per_nfe_summary.txt
