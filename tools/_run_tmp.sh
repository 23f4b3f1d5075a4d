ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbstamp.so python tools/rb_stamps.py 27 1 2>&1 | tail -7
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "every_layer_bf16 or batch8 or class_cond or cfg" 2>&1 | tail -3 | cut -c1-600
bash tools/ab_layers.sh ADF_GEMM_RB 1 > gpurun_out/ab_layers.txt 2>&1; grep -E "rb ?(0|4|23|25|27) |total" gpurun_out/ab_layers.txt
bash tools/ab_bench.sh ADF_GEMM_RB 1
