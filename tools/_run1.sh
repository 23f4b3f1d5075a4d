set -e
cd $GRAFT_REPO_ROOT
B="--steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads"
for i in 1 2 3; do
for v in base "" slp; do
  if [ -z "$v" ]; then lib=""; else lib=audiodiffuser_amd/build/variants/libadf_hip_$v.so; fi
  ms=$(ADF_HIP_LIB=$lib timeout -k 10 200 python bench.py $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],1), round(d["roofline"]["ms_per_launch"]*1000,1), round(d["roofline"]["all_resblocks"]["ms"],3))')
  echo "variant=${v:-product} ms_per_step, dominant us, all resblocks ms = $ms"
done; done
echo "== slp parity"; ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_slp.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resblock or conv or bf16 or route" 2>&1 | tail -3
