#!/bin/bash
# SQ counters of one bench.py workload (kernel trace + one --pmc pass per set, nothing else), summarised per kernel:
#   tools/sq_counters.sh <tag> <kernel-name grep> -- <bench.py args>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
tag=$1; kgrep=$2; shift 3
O=$R/gpurun_out/r06_sq_$tag; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  t=$(echo $set | cut -d' ' -f1)
  (cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$t -o p -- python $R/bench.py --no-cpu-baseline --no-secondary --no-parity-check --steps 4 --warmup 1 --full-json - "$@" > $O/pmc_$t.log 2>&1)
  python tools/pmc_summary.py $O/pmc_$t 2>&1 | grep -A10 "$kgrep" | head -12
  rm -rf $O/pmc_$t
done 2>&1 | tee $O/sq_$tag.txt
