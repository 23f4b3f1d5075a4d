#!/bin/bash
# One gpurun call: bf16 parity report, the GPU test-suite, a bench line.  Logs under gpurun_out/.
mkdir -p gpurun_out
python tests/diag/gpu_bf16_parity_report.py "$@" > gpurun_out/parity_report.log 2>&1; echo "parity report rc=$?"
tail -12 gpurun_out/parity_report.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gputests.log 2>&1; echo "gpu tests rc=$?"
tail -25 gpurun_out/gputests.log
timeout -k 10 400 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
