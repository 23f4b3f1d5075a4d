"""Summarise a rocprofv3 kernel_trace.csv: per-kernel totals inside the sampler window (per NFE)."""
import csv, collections, sys, glob
path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof_*/**/*kernel_trace.csv', recursive=True))[-1]
nfe = int(sys.argv[2]) if len(sys.argv) > 2 else 99
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
first = next(i for i, n in enumerate(names) if 'edm_coef' in n)
last = max(i for i, n in enumerate(names) if 'rk2_kernel' in n or 'euler_kernel' in n or 'dpm_kernel' in n)
sel = rows[first:last + 1]
wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r['Kernel_Name'].replace('adf::', '').replace('void ', '').split('(')[0]
    k += f" g{r['Grid_Size_X']}" if '--grid' in sys.argv else ''
    agg[k][0] += 1; agg[k][1] += d
busy = sum(v[1] for v in agg.values()) / 1e3
print(f"{path}\nsampler window: wall {wall:.1f} ms, kernel busy {busy:.1f} ms, launches {len(sel)}, per NFE {wall/nfe:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:62s} n/NFE={v[0]/nfe:6.1f} ms/NFE={v[1]/1e3/nfe:7.3f} avg={v[1]/v[0]:7.1f}us {v[1]/1e3/busy*100:5.1f}%")
