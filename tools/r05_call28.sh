#!/bin/bash
# Round 5, call 28: k_fir_ring's chains paced by wave PRIORITY (an early run runs at priority 0 until its lead is up: the
# SIMD's other wave takes the issue slots, nobody sleeps) against waits and against free-running chains.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ab
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 240 python tools/fir_map_probe.py > $O/probe_exact.log 2> $O/probe_exact.err; echo "probe rc=$?"; cut -c1-250 $O/probe_exact.log
timeout 200 python tools/fir_map_probe.py --fused 1 > $O/probe_fma.log 2> $O/probe_fma.err; echo "probe fma rc=$?"; cut -c1-250 $O/probe_fma.log
for shape in "4096 524288 256" "16384 131072 256" "32768 65536 256"; do
  set -- $shape
  for f in 0 1; do
    timeout 200 python tools/fir_map_probe.py --configs map1,free,prio_h0 --channels $1 --rows $2 --taps $3 --fused $f 2>> $O/probe_shapes.err | sed "s/^{/{\"channels\": $1, \"rows\": $2, /" | tee -a $O/probe_shapes.log | cut -c1-270
  done
done
pmc() {  # key, probe args
  key=$1; shift
  cd /tmp
  ALZ_FIR_WAITSTAT=1 timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw_$key -o p -- python $R/tools/fir_map_probe.py "$@" > $O/raw_$key.log 2>&1
  python $R/tools/pmc_sum.py $O/raw_$key FETCH_SIZE 3 > $O/pmc_$key.json 2>> $O/errors.log
  rm -rf $O/raw_$key
  echo "$key: $(cut -c1-120 $O/pmc_$key.json) $(grep 'fir chains' $O/raw_$key.log | tail -1)"
}
for c in prio prio_h0 prio_s100_h0 prio_s70_h0 prio_s120_h0; do pmc $c --only $c; done
pmc prio_h0_fma --only prio_h0 --fused 1
for shape in "4096 524288 256" "16384 131072 256" "32768 65536 256"; do
  set -- $shape
  pmc prio_h0_c$1 --only prio_h0 --channels $1 --rows $2 --taps $3
done
