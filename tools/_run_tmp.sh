ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbphs.so timeout -k 10 120 python tools/rb_stamps.py 27 1 2>&1 | grep -E "^rc|^T" | head -12
for v in "" audiodiffuser_amd/build/variants/libadf_hip_rbph.so; do ADF_HIP_LIB=$v ADF_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --roofline-only --roofline-iters 30 2>/dev/null | tail -1 | python3 -c "
import json,sys
rows=json.loads(sys.stdin.read())['rows']
print(' '.join('%d.%d:%.1f'%(r['resblock'],r['kernel'],r['ms']*1e3) for r in rows if r['resblock'] in (0,2,4,23,25,27)), 'total %.1f'%(sum(r['ms'] for r in rows)*1e3))"; done
for v in "" audiodiffuser_amd/build/variants/libadf_hip_rbph.so; do ADF_HIP_LIB=$v timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-precision-check 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done
