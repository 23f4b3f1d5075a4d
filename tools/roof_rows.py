import json, sys
r = json.load(open(sys.argv[1]))['roofline']
sel = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else None
tot = 0
for row in r['rows']:
    tot += row['ms']
    if sel and row['resblock'] not in sel: continue
    if row['ms'] <= 0:                       # fused short-level block: no separate conv launches to replay
        print(f"rb{row['resblock']:2d} k{row['kernel']}   (one fused launch)", end='   ' if row['kernel'] == 1 else '\n')
        continue
    tf = row['flops'] / row['ms'] / 1e9; gb = row['bytes'] / row['ms'] / 1e6
    print(f"rb{row['resblock']:2d} k{row['kernel']} {row['ms']*1e3:7.1f}us {tf:6.1f}TF {gb:7.1f}GB/s", end='   ' if row['kernel'] == 1 else '\n')
print('total ms', round(tot, 3))
