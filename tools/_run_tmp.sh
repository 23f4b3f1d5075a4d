ADF_RB_FUSED=0 ADF_TR_FUSED=0 python tests/diag/gpu_forced_report.py c3 2 4096 2>/dev/null | tail -1 > gpurun_out/forced_unfused.json
python3 - <<'PY'
import json
f=json.load(open('gpurun_out/forced_unfused.json'))['forced']
import collections
cls=collections.defaultdict(list)
for k,v in f.items():
    suf=k.split('.')[-1]
    key = suf if '.attn.' in k or suf in ('h1','attn','conv') else ('block' if 'block' in k or k.startswith('mid.') else k)
    cls[key].append(v)
for k,v in cls.items(): print(k, len(v), "max %.2e"%max(v))
PY
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "every_layer_bf16 or unfused or attention_at_1024 or config3_every" 2>&1 | tail -4 | cut -c1-500
