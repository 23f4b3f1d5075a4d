#!/bin/bash
# correctness of the rb kernel against the pp / ws routes (same bf16 net, B=8 with ADF_GEMM_RB=2 and B=24), then layer timings
mkdir -p gpurun_out
for B in 8 24; do
  ADF_GEMM_RB=0 B=$B timeout -k 10 200 python tests/diag/gpu_pp_check.py save /tmp/rb0_$B.pt || exit 1
  ADF_GEMM_RB=2 B=$B timeout -k 10 200 python tests/diag/gpu_pp_check.py save /tmp/rb2_$B.pt || exit 1
  echo "--- B=$B rb vs other routes"
  VERBOSE=1 python tests/diag/gpu_pp_check.py cmp /tmp/rb2_$B.pt /tmp/rb0_$B.pt > gpurun_out/rb_cmp_$B.txt 2>&1
  tail -3 gpurun_out/rb_cmp_$B.txt
  awk '$4+0 > 3e-3 || $NF+0 > 0' gpurun_out/rb_cmp_$B.txt | head -20
done
bash tools/ab_layers.sh ADF_GEMM_RB 0 1 > gpurun_out/ab_layers.txt 2>&1
paste <(grep -A100 "RB=0" gpurun_out/ab_layers.txt | grep -B100 "RB=1" | grep "^rb\|^total") <(grep -A100 "RB=1" gpurun_out/ab_layers.txt | grep "^rb\|^total") | head -70
bash tools/ab_bench.sh ADF_GEMM_RB 0 1
