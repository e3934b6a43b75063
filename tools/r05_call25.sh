#!/bin/bash
# Round 5, call 25: the paced chains of k_fir_ring -- member rotation (so that the two waves of a SIMD do not wait
# together), bounded waits, chain widths 4 / 8 / 16: time, wait statistics and FETCH_SIZE.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05y
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
LIST="chain_rot2 chain_b100 chain_b30 chain_w8_pct67 chain_w8_pct67_rot4 chain_w16_pct33 chain_w16_pct33_b100 chain_w16_pct33_b300 chain_w16_pct20"
CF=map1,chain_noflags,$(echo $LIST | tr ' ' ',')
timeout 240 python tools/fir_map_probe.py --configs $CF > $O/probe_exact.log 2> $O/probe_exact.err; echo "probe rc=$?"; cut -c1-250 $O/probe_exact.log
timeout 200 python tools/fir_map_probe.py --fused 1 --configs $CF > $O/probe_fma.log 2> $O/probe_fma.err; echo "probe fma rc=$?"; cut -c1-250 $O/probe_fma.log
pmc() {  # key, probe args
  key=$1; shift
  cd /tmp
  ALZ_FIR_WAITSTAT=1 timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw_$key -o p -- python $R/tools/fir_map_probe.py "$@" > $O/raw_$key.log 2>&1
  python $R/tools/pmc_sum.py $O/raw_$key FETCH_SIZE 3 > $O/pmc_$key.json 2>> $O/errors.log
  rm -rf $O/raw_$key
  echo "$key: $(cut -c1-120 $O/pmc_$key.json) $(grep 'fir chains' $O/raw_$key.log | tail -1)"
}
for c in $LIST; do pmc $c --only $c; done
