#!/bin/bash
# round 4, call 21: attention kernel with the probability row sums on the matrix cores: parity (1-D nets, ADM head dim 64), timing of the N = 1024 launches, c3 step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c21; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_adm.py -x -q -m gpu -k "attention or attn or unfused or every_layer_bf16 or c3 or adm" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
rm -rf /tmp/pa /tmp/pt
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex attention_mfma32 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pa -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-pmc --no-precision-check > /tmp/pa.log 2>&1 || { tail -5 /tmp/pa.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-pmc --no-precision-check > /tmp/pt.log 2>&1 && python3 tools/attention_util.py /tmp/pa /tmp/pt > $out/attention_mfma_utilisation.txt
cat $out/attention_mfma_utilisation.txt
timeout -k 10 400 python3 bench.py --config c3 --sampler dpm --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > $out/bench_c3.json 2> $out/bench_c3.err || { tail -5 $out/bench_c3.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$out/bench_c3.json').read().strip().splitlines()[-1]); print('c3', d['ms_per_step'])"
