cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "config3_every or unfused or every_layer_bf16 or c1_full or test_net_and" 2>&1 | tail -3 | cut -c1-600
rm -rf /tmp/p1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-precision-check > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
grep ms_per_step /tmp/p1.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('c3 dpm ms_per_step', j['ms_per_step'])"
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 98 --grid | grep -E "attention|to_in|to_out|window" 
