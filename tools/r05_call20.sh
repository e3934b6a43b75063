#!/bin/bash
# Round 5, call 20: run-to-wave mappings of k_fir_ring (verdict item 7: FETCH_SIZE of configs[2] is 2.9 x its input).
# tools/fir_map_probe.py on the -DALZ_TUNING build: bitwise equality with the shipped mapping, time, then FETCH_SIZE per
# launch for the candidates (separate --pmc passes, kernel trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05t
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 240 python tools/fir_map_probe.py > $O/probe_exact.log 2> $O/probe_exact.err; echo "probe rc=$?"; cat $O/probe_exact.log | cut -c1-230
timeout 200 python tools/fir_map_probe.py --fused 1 > $O/probe_fma.log 2> $O/probe_fma.err; echo "probe fma rc=$?"; cat $O/probe_fma.log | cut -c1-230
for cfg in map1 cohort cohort_stag cohort_stag_nty map1_nty cohort_stag_nty_fill16 cohort_stag_nty_g8; do
  cd /tmp
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw_$cfg -o p -- python $R/tools/fir_map_probe.py --only $cfg > $O/raw_$cfg.log 2>&1
  python $R/tools/pmc_sum.py $O/raw_$cfg FETCH_SIZE 3 > $O/pmc_$cfg.json 2>> $O/errors.log
  rm -rf $O/raw_$cfg
  echo "$cfg: $(cut -c1-260 $O/pmc_$cfg.json)"
done
