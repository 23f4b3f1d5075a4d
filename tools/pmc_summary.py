"""Per-dispatch PMC summary for conv_gemm launches of the resblock replay (bench.py --roofline-only)."""
import csv, glob, collections, sys
def load(d):
    f = glob.glob(f'{d}/*/*counter_collection.csv')[0]
    rows = list(csv.DictReader(open(f)))
    byd = collections.OrderedDict()
    for r in rows:
        k = r['Dispatch_Id']
        e = byd.setdefault(k, {'name': r['Kernel_Name'], 'grid': r['Grid_Size'], 'wg': r['Workgroup_Size'], 'lds': r.get('LDS_Block_Size'), 'vgpr': r.get('VGPR_Count')})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    return byd
out = {}
for d in sys.argv[1:]:
    for k, e in load(d).items():
        if 'conv_gemm' not in e['name']: continue
        nm = e['name']
        key = ((nm.split('(')[0].split('::')[-1].split('<')[0] + ' ' + (nm.split('<')[1].split('>')[0] if '<' in nm else '')).strip(), e['grid'])
        o = out.setdefault(key, collections.defaultdict(list))
        for c, v in e.items():
            if isinstance(v, float): o[c].append(v)
for key, o in out.items():
    import os
    if int(key[1]) < int(os.environ.get("PMC_MIN_GRID", "200000")): continue
    print(key, {c: round(sum(v[-3:]) / len(v[-3:]), 1) for c, v in o.items()}, 'n=', max(len(v) for v in o.values()))
