"""List VGPR/AGPR/SGPR/LDS/spill figures per kernel from a hipcc -S assembly file (usage: kernel_resources.py file.s)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
md = txt[txt.find('amdhsa.kernels'):]
for blk in md.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, ''])[1]
    name = g('name')
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        dn = name
    dn = re.sub(r'\(.*', '', dn).replace('void adf::', '')
    print(f"{dn:64s} agpr {blk.split()[0]:>4s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>7s} "
          f"spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s}")
