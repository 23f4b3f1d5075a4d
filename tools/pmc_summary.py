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

# distinct (FETCH_SIZE, WRITE_SIZE) pairs of the persistent DMA kernel, one per layer shape (PMC_PAIRS=1)
import os
if os.environ.get("PMC_PAIRS") and len(sys.argv) >= 3:
    dma = lambda e: 'pp_kernel' in e['name'] or 'rb_kernel' in e['name']
    f = {k: e for k, e in load(sys.argv[1]).items() if dma(e)}
    w = {k: e for k, e in load(sys.argv[2]).items() if dma(e)}
    pairs = collections.Counter()
    for (kf, ef), (kw, ew) in zip(f.items(), w.items()):      # same launch order in both passes
        pairs[(round(ef.get('FETCH_SIZE', 0) / 256) * 256, round(ew.get('WRITE_SIZE', 0) / 256) * 256)] += 1
    print("# conv_gemm_pp_kernel / conv_gemm_rb_kernel dispatches by (FETCH_SIZE KB, WRITE_SIZE KB), rounded to 256 KB: count")
    for k, v in sorted(pairs.items()):
        print(k, v, " HBM bytes/launch = 2*FETCH + WRITE = %.1f MB" % ((2 * k[0] + k[1]) * 1024 / 1e6))
