#!/bin/bash
# Is the dense Levinson-Durbin tail of k_acorr_stage<17, 2> bound by instruction fetch?  SQ / SQC counters of the 65 536-frame launch
# through three builds of alz_lev.h (tools/variants/libalzhip_<lib>.so; "shipped" = the library in the tree).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06_lpcpmc
mkdir -p $O
B="--workload lpc --lpc-exact --no-cpu-baseline --no-secondary --no-parity-check --steps 6 --warmup 2 --full-json -"
for lib in ${LIBS:-lev_old lev_nopipe}; do
  if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_LEVEL_VMEM SQ_WAVES"; do
    tag=${lib}_$(echo $set | cut -d' ' -f1)
    (cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $R/bench.py $B > $O/pmc_$tag.log 2>&1)
    echo "== $lib"; python tools/pmc_summary.py $O/pmc_$tag 2>&1 | grep -A10 "k_acorr_stage" | head -12
    rm -rf $O/pmc_$tag
  done
done 2>&1 | tee $O/lpc_ifetch_pmc.txt
