#!/bin/bash
# one iteration of the rb kernel work: stamps of one K block, correctness vs the other routes at B=8, per-layer replay, bench
mkdir -p gpurun_out
#ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbstamp.so timeout -k 10 120 python tools/rb_stamps.py 27 1 2>&1 | tail -13
ADF_GEMM_RB=0 B=8 timeout -k 10 200 python tests/diag/gpu_pp_check.py save /tmp/rb0.pt > /dev/null || exit 1
ADF_GEMM_RB=2 B=8 timeout -k 10 200 python tests/diag/gpu_pp_check.py save /tmp/rb2.pt > /dev/null || exit 1
VERBOSE=1 python tests/diag/gpu_pp_check.py cmp /tmp/rb2.pt /tmp/rb0.pt > gpurun_out/rb_cmp.txt 2>&1; grep -E "down0.conv |down1.conv |down2.conv |down0.block0 |down2.block0 |up5.block1 |^worst" gpurun_out/rb_cmp.txt
bash tools/ab_layers.sh ADF_GEMM_RB 1 > gpurun_out/ab_layers.txt 2>&1; grep -E "rb ?(0|4|23|25|27) |total" gpurun_out/ab_layers.txt
bash tools/ab_bench.sh ADF_GEMM_RB 0 1
