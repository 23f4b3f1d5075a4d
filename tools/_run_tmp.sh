timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "sampler or golden or graph or cfg or dpm or lms or error or nfe or injected" 2>&1 | tail -3 | cut -c1-600
bash tools/ab_bench.sh ADF_GEMM_RB 1
