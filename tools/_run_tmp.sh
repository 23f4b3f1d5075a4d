mkdir -p gpurun_out
python tests/diag/gpu_bf16_parity_report.py > gpurun_out/parity_report.log 2>&1; tail -9 gpurun_out/parity_report.log | cut -c1-200
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/gputests.log | cut -c1-600
