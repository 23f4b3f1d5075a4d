"""MFMA utilisation of attention_mfma32_kernel at 1024 tokens (BASELINE configs[2] net) from (a) a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES,
SQ_INSTS_MFMA, SQ_INSTS_VALU per dispatch) and (b) a plain --kernel-trace pass (durations WITHOUT counter collection).
utilisation = MFMA-busy SIMD-cycles / (duration x clock x 1024 SIMDs).  Round 3 divided by GRBM_GUI_ACTIVE, which the guide says reads high on dispatches
shorter than ~0.3 ms (MI355X_MICROARCH.md "DVFS give-back"): that is where its 13 % came from.
usage: python tools/attention_util.py DIR_PMC DIR_TRACE"""
import collections, csv, glob, sys
pmc = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
trc = glob.glob(f"{sys.argv[2]}/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(pmc)):
    if "attention_mfma32" in r["Kernel_Name"]:
        agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trc)):
    if "attention_mfma32" in r["Kernel_Name"]:
        dur[r.get("Grid_Size_X") or r.get("Grid_Size")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g, c in sorted(agg.items(), key=lambda kv: -int(kv[0])):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    d = dur.get(g, [])
    if not d:
        continue
    us = sorted(d)[len(d) // 2]
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    line = f"grid {g}: {len(d)} launches, median {us:.1f} us without counters; per launch: MFMA {m.get('SQ_INSTS_MFMA', 0):,.0f}, VALU {m.get('SQ_INSTS_VALU', 0):,.0f}"
    if m.get("SQ_INSTS_MFMA"):
        line += f" ({m.get('SQ_INSTS_VALU', 0) / m['SQ_INSTS_MFMA']:.1f} per MFMA)"
    print(line)
    for ghz in (2.4, 2.0):
        print(f"    MFMA-busy {busy:,.0f} SIMD-cycles / ({us:.1f} us x {ghz} GHz x 1024 SIMDs) = {100.0 * busy / (us * 1e-6 * ghz * 1e9 * 1024):.1f} % MFMA utilisation at {ghz} GHz")
