#!/bin/bash
# Runs ON the GPU box (gpurun): produces the small summaries that get committed under profiles/ (named per round).
# usage: tools/collect_profiles.sh r01      -> gpurun_out/profiles_r01/*
set -o pipefail
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "[1/5] bench.py"; timeout -k 10 400 python bench.py --steps 2 --warmup 1 2>/dev/null | tail -1 > $out/${tag}_bench.json || exit 1
echo "[2/5] kernel trace of bench.py"; rm -rf /tmp/p1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
cp $(ls /tmp/p1/*/*kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 198 --grid | sed "s#/tmp/p1/[^ ]*#rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline#" > $out/${tag}_bench_per_nfe_summary.txt
echo "[3/5] kernel trace of the resblock replay"; rm -rf /tmp/p2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 bench.py --roofline-only --roofline-iters 50 > /tmp/p2.log 2>&1 || { tail -5 /tmp/p2.log; exit 1; }
cp $(ls /tmp/p2/*/*kernel_stats.csv | head -1) $out/${tag}_roofline_replay_kernel_stats.csv
ADF_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --roofline-only --roofline-iters 50 2>/dev/null | tail -1 > $out/${tag}_roofline_replay.json
echo "[4/5] PMC FETCH_SIZE (resblock replay, conv_gemm kernels)"; rm -rf /tmp/p3 /tmp/p4
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "conv_gemm" --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -- python3 bench.py --roofline-only --roofline-iters 2 > /tmp/p3.log 2>&1 || { tail -5 /tmp/p3.log; exit 1; }
echo "[5/5] PMC WRITE_SIZE"
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "conv_gemm" --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -- python3 bench.py --roofline-only --roofline-iters 2 > /tmp/p4.log 2>&1 || { tail -5 /tmp/p4.log; exit 1; }
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --roofline-only --roofline-iters 2"
  echo "# per kernel variant and grid: mean of the last 3 dispatches, KB as reported (gfx950: double FETCH_SIZE for 16 B/lane streaming reads)"
  PMC_MIN_GRID=1 PMC_PAIRS=1 python3 tools/pmc_summary.py /tmp/p3 /tmp/p4; } > $out/${tag}_pmc_fetch_write_summary.txt
ls -la $out
