#!/bin/bash
# round 4, call 7: the chained short-level resblock kernel: route test (bit for bit against the two-launch form), B = 64 parity, A/B
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c7; mkdir -p $out
echo "== route test + parity"
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_short_level or config2_batch64_full_length or batch64_bf16_is_batch_independent or graph_replay_is_repeatable or bf16_heun" 2>&1 | tail -8 || exit 1
ab() {
  ms=$(env $1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), d["device_loop"])')
  echo "$1: $ms"
}
for rep in 1 2 3; do ab ADF_RB_CHAIN=1; ab ADF_RB_CHAIN=0; done | tee $out/ab.txt
echo "== per-NFE kernel table"; rm -rf /tmp/p1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 198 --grid > $out/bench_per_nfe_summary.txt; head -36 $out/bench_per_nfe_summary.txt
