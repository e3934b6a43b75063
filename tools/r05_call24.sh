#!/bin/bash
# Round 5, call 24: where the paced chains of k_fir_ring lose their 9 %: wait statistics (-DALZ_TUNING build,
# ALZ_FIR_WAITSTAT), then time and FETCH_SIZE of the one-chain-per-group variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05x
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
for c in chain chain_pct0 chain_fill64 chain_w16_pct33 chain_w16_pct25 chain_w16_pct33_fill64; do
  echo "== $c"; ALZ_FIR_WAITSTAT=1 timeout 100 python tools/fir_map_probe.py --only $c 2>&1 | grep "fir chains"
  echo "== $c fma"; ALZ_FIR_WAITSTAT=1 timeout 100 python tools/fir_map_probe.py --only $c --fused 1 2>&1 | grep "fir chains" | tail -1
done 2>&1 | tee $O/waitstat.log
CF=map1,chain_noflags,chain_w16_pct33,chain_w16_pct25,chain_w16_pct33_fill64,chain_w16_pct33_fill32
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
for c in chain_w16_pct25 chain_w16_pct33_fill64 chain_w16_pct33_fill32; do pmc $c --only $c; done
