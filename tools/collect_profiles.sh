#!/bin/bash
# Runs ON the GPU box (gpurun): produces the small summaries that get committed under profiles/ (named per round).
# usage: tools/collect_profiles.sh r01      -> gpurun_out/profiles_r01/*
set -o pipefail
tag=${1:-r04}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "[1/9] bench.py"; timeout -k 10 400 python bench.py --steps 2 --warmup 1 2>/dev/null | tail -1 > $out/${tag}_bench.json || exit 1
echo "[2/9] kernel trace of bench.py"; rm -rf /tmp/p1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
cp $(ls /tmp/p1/*/*kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
python3 tools/trace_summary.py $(ls /tmp/p1/*/*kernel_trace.csv | head -1) 198 --grid | sed "s#/tmp/p1/[^ ]*#rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc#" > $out/${tag}_bench_per_nfe_summary.txt
echo "[3/9] kernel trace of the resblock replay"; rm -rf /tmp/p2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 bench.py --roofline-only --roofline-iters 50 > /tmp/p2.log 2>&1 || { tail -5 /tmp/p2.log; exit 1; }
cp $(ls /tmp/p2/*/*kernel_stats.csv | head -1) $out/${tag}_roofline_replay_kernel_stats.csv
ADF_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --roofline-only --roofline-iters 50 2>/dev/null | tail -1 > $out/${tag}_roofline_replay.json
echo "[4/9] PMC FETCH_SIZE (resblock replay, conv_gemm kernels)"; rm -rf /tmp/p3 /tmp/p4
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "conv_gemm" --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -- python3 bench.py --roofline-only --roofline-iters 2 > /tmp/p3.log 2>&1 || { tail -5 /tmp/p3.log; exit 1; }
echo "[5/9] PMC WRITE_SIZE"
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "conv_gemm" --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -- python3 bench.py --roofline-only --roofline-iters 2 > /tmp/p4.log 2>&1 || { tail -5 /tmp/p4.log; exit 1; }
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --roofline-only --roofline-iters 2"
  echo "# per kernel variant and grid: mean of the last 3 dispatches, KB as reported (gfx950: double FETCH_SIZE for 16 B/lane streaming reads)"
  PMC_MIN_GRID=1 PMC_PAIRS=1 python3 tools/pmc_summary.py /tmp/p3 /tmp/p4; } > $out/${tag}_pmc_fetch_write_summary.txt
echo "[6/9] per-launch layer table (route lines matched with the kernel trace)"; rm -rf /tmp/lt
ADF_GEMM_TRACE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 bench.py --steps 1 --warmup 0 --num-steps 2 --no-graph --no-cpu-baseline --no-pmc --no-precision-check --no-other-workloads > /tmp/lt.log 2> /tmp/lt.err || { tail -5 /tmp/lt.err; exit 1; }
{ echo "# ADF_GEMM_TRACE=1 rocprofv3 --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --num-steps 2 --no-graph (C2, B = 64, L = 16384, bf16)"
  python3 tools/layer_table.py /tmp/lt /tmp/lt.err; } > $out/${tag}_layer_table.txt || exit 1
echo "[7/9] SQ counters of the rb kernel (resblock replay)"
rm -rf gpurun_out/pmc_sq; bash tools/pmc_sq.sh gpurun_out/pmc_sq rb_kernel > /tmp/sq.log 2>&1; cp gpurun_out/pmc_sq/sq_counters.txt $out/${tag}_rb_kernel_sq_counters.txt
echo "[8/9] attention PMC (C3, DPM)"; rm -rf /tmp/pa
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex attention_mfma32 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pa -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-pmc --no-precision-check > /tmp/pa.log 2>&1 || { tail -5 /tmp/pa.log; exit 1; }
python3 - > $out/${tag}_attention_pmc_raw.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pa/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for g, c in sorted(agg.items(), key=lambda kv: -int(kv[0])):
    print('grid', g, ' '.join(f"{k} {sum(v)/len(v):,.0f}" for k, v in c.items()), ' n=%d' % max(len(v) for v in c.values()))
t = glob.glob('/tmp/pa/*/*kernel_trace.csv')
if t:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(t[0])):
        d[r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for g, v in sorted(d.items(), key=lambda kv: -int(kv[0])):
        print(f"grid_x {g}: {len(v)} launches, mean {sum(v)/len(v):.1f} us (under counter collection)")
PY
rm -rf /tmp/pt
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 bench.py --config c3 --sampler dpm --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-pmc --no-precision-check > /tmp/pt.log 2>&1 && python3 tools/attention_util.py /tmp/pa /tmp/pt > $out/${tag}_attention_mfma_utilisation.txt
echo "[9/9] C3 bench line"
timeout -k 10 400 python bench.py --config c3 --sampler dpm --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${tag}_bench_c3_dpm.json
echo "[10] WaveNet (config 5): bench line, kernel trace, parity report, phase stamps"
timeout -k 10 600 python bench.py --config c5 --steps 1 --warmup 1 2>/dev/null | tail -1 > $out/${tag}_bench_c5_wavenet.json
rm -rf /tmp/p5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python3 bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-precision-check > /tmp/p5.log 2>&1 || { tail -5 /tmp/p5.log; exit 1; }
cp $(ls /tmp/p5/*/*kernel_stats.csv | head -1) $out/${tag}_bench_c5_kernel_stats.csv
for w in 1 0; do ADF_WN_WIDE=$w timeout -k 10 300 python tests/diag/gpu_wn_report.py 4200 2 > /dev/null 2>&1; done
cp gpurun_out/wn_parity_report_wide1.json $out/${tag}_wavenet_bf16_parity_wide.json; cp gpurun_out/wn_parity_report_wide0.json $out/${tag}_wavenet_bf16_parity_64.json
if [ -f audiodiffuser_amd/build/variants/libadf_hip_wnstamp.so ]; then
  { for l in 0 5 11 35; do ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_wnstamp.so timeout -k 10 200 python tools/wn_stamps.py $l 128 2>&1 | grep -v amdgpu.ids; done
    echo "# --- 64-position route (ADF_WN_WIDE=0)"
    ADF_WN_WIDE=0 ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_wnstamp.so timeout -k 10 200 python tools/wn_stamps.py 5 128 2>&1 | grep -v amdgpu.ids; } > /tmp/wn_stamps.txt
  # (a stale variant build fails to load: round 2's committed file was once overwritten by such a traceback -- keep the output only when every run reported rc 0)
  if grep -q Traceback /tmp/wn_stamps.txt; then echo "wn_stamps failed (stale variant build?): not kept"; else cp /tmp/wn_stamps.txt $out/${tag}_wavenet_layer_stamps_collection.txt; fi
fi
echo "[11] ADM 2-D U-Net (config 4): bench line, kernel trace"
timeout -k 10 600 python bench.py --config c4 --steps 1 --warmup 1 2>/dev/null | tail -1 > $out/${tag}_bench_c4_adm.json
rm -rf /tmp/p6
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6 -- python3 bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-precision-check > /tmp/p6.log 2>&1 || { tail -5 /tmp/p6.log; exit 1; }
cp $(ls /tmp/p6/*/*kernel_stats.csv | head -1) $out/${tag}_bench_c4_kernel_stats.csv
rm -rf /tmp/p7
( cd /tmp && ADF_C2_TRACE=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p7 -- python3 $OLDPWD/tools/adm_pass.py 64 2> /tmp/p7.err > /dev/null ) || { tail -5 /tmp/p7.err; exit 1; }
python3 tools/adm_layer_table.py /tmp/p7 /tmp/p7.err > $out/${tag}_adm_layer_table.txt
echo "[12] whole-launch timelines of the resblock conv kernel (diagnostic build) and the SIMD issue microbenchmark"
if [ -f audiodiffuser_amd/build/variants/libadf_hip_rbtl.so ]; then
  ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbtl.so timeout -k 10 200 python tools/rb_timeline.py 28 1 2>&1 | grep -v amdgpu.ids > $out/${tag}_rb_launch_timeline.txt
  ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbtl.so timeout -k 10 200 python tools/rb_timeline.py 4 1 2>&1 | grep -v amdgpu.ids > $out/${tag}_rb_launch_timeline_n256.txt
  ADF_HIP_LIB=audiodiffuser_amd/build/variants/libadf_hip_rbtl.so timeout -k 10 200 python tools/rb_timeline.py 20 1 64 single 2>&1 | grep -v amdgpu.ids > $out/${tag}_rb_launch_timeline_L256.txt
fi
if [ -x tools/micro/bin/rb_floor ]; then timeout -k 5 200 tools/micro/bin/rb_floor > $out/${tag}_rb_floor_microbench.txt 2>&1; fi
echo "[13] split-bf16 mode: per-kernel table"; rm -rf /tmp/p8
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p8 -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads > /tmp/p8.log 2>&1 && python3 tools/trace_summary.py $(ls /tmp/p8/*/*kernel_trace.csv | head -1) 198 --grid | sed "s#/tmp/p8/[^ ]*#rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype f32x3 --steps 1 --warmup 1#" > $out/${tag}_f32x3_mode_per_nfe_summary.txt
if [ -x tools/micro/bin/role_split ] && false; then timeout -k 5 100 tools/micro/bin/role_split > $out/${tag}_role_split_microbench_raw.txt 2>&1; fi
ls -la $out
