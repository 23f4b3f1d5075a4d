#!/bin/bash
# Runs ON the GPU box: SQ / SQC counters of the DMA kernel launches of the resblock replay, one rocprofv3 pass per
# counter group (a pass with an unknown counter name fails alone).  usage: tools/pmc_sq.sh outdir [kernel regex]
out=${1:-gpurun_out/pmc_sq}; mkdir -p $out
kre=${2:-rb_kernel}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1)); rm -rf /tmp/pq$i
  echo "[pass $i] $grp"
  timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex "$kre" --pmc $grp --output-format csv -d /tmp/pq$i -- python3 bench.py --roofline-only --roofline-iters 2 > /tmp/pq$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pq$i.log; continue; }
  PMC_MIN_GRID=1 python3 tools/pmc_summary.py /tmp/pq$i | grep "131072" >> $out/sq_counters.txt
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
GROUPS
cat $out/sq_counters.txt
