#!/bin/bash
# Round 5, call 23: k_fir_ring in CHAINS (runs walked towards the past, run r started when run r + 1 is R taps in):
# bitwise equality with the shipped mapping, time, FETCH_SIZE per launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05w
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
CF=map1,chain,chain_nty,chain_noflags,chain_pct50,chain_pct75,chain_pct125,chain_fill64,chain_fill32,chain_w8_pct50,chain_w16_pct33
timeout 240 python tools/fir_map_probe.py --configs $CF > $O/probe_exact.log 2> $O/probe_exact.err; echo "probe rc=$?"; cut -c1-250 $O/probe_exact.log
timeout 200 python tools/fir_map_probe.py --fused 1 --configs $CF > $O/probe_fma.log 2> $O/probe_fma.err; echo "probe fma rc=$?"; cut -c1-250 $O/probe_fma.log
pmc() {  # key, probe args
  key=$1; shift
  cd /tmp
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw_$key -o p -- python $R/tools/fir_map_probe.py "$@" > $O/raw_$key.log 2>&1
  python $R/tools/pmc_sum.py $O/raw_$key FETCH_SIZE 3 > $O/pmc_$key.json 2>> $O/errors.log
  rm -rf $O/raw_$key
  echo "$key: $(cut -c1-200 $O/pmc_$key.json)"
}
for c in chain chain_pct50 chain_pct75 chain_pct125 chain_fill64 chain_fill32 chain_w8_pct50 chain_w16_pct33; do pmc $c --only $c; done
pmc chain_fma --only chain --fused 1
