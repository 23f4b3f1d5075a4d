#!/bin/bash
# round 4, call 6: is the step power-limited?  socket power and clocks while bench.py runs; rb_floor with the 16x16x32 MFMA shape
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c6; mkdir -p $out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30 | tee $out/smi_idle.txt
echo "== rb_floor"; timeout -k 10 200 tools/micro/bin/rb_floor | tee $out/rb_floor.txt
echo "== bench with power samples"
bash tools/power_probe.sh $out/power_bench.txt -- timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-precision-check --no-pmc --no-other-workloads 2>/dev/null | tail -1 | cut -c1-200
grep -c t= $out/power_bench.txt; awk 'NR%8==0' $out/power_bench.txt | cut -c1-400 | head -40
