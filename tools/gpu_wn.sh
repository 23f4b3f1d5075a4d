#!/bin/bash
# One gpurun call for the WaveNet path: parity tests, then the c5 bench line (small first, then the BASELINE size).
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_wavenet.py -m gpu -x -q > gpurun_out/wn_tests.log 2>&1; rc=$?; echo "wn tests rc=$rc"; tail -5 gpurun_out/wn_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --config c5 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/bench_c5_b16.json 2> gpurun_out/bench_c5_b16.err; rc=$?; echo "c5 b16 rc=$rc"
tail -c 2500 gpurun_out/bench_c5_b16.json; tail -3 gpurun_out/bench_c5_b16.err
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --config c5 --steps 1 --warmup 1 "$@" > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; echo "c5 rc=$?"
tail -c 4000 gpurun_out/bench_c5.json; tail -3 gpurun_out/bench_c5.err
