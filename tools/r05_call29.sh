#!/bin/bash
# Round 5, call 29: k_fir_ring's chains with BOUNDED waits (a run waits for its lead at most `bound` percent of the lead)
# and chain widths 4 / 8 / all resident waves: time, wait statistics, FETCH_SIZE; configs[2] and other bank widths.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ac
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 240 python tools/fir_map_probe.py > $O/probe_exact.log 2> $O/probe_exact.err; echo "probe rc=$?"; cut -c1-250 $O/probe_exact.log
timeout 200 python tools/fir_map_probe.py --fused 1 > $O/probe_fma.log 2> $O/probe_fma.err; echo "probe fma rc=$?"; cut -c1-250 $O/probe_fma.log
for shape in "4096 524288 256" "16384 131072 256" "32768 65536 256" "8192 262144 128"; do
  set -- $shape
  for f in 0 1; do
    timeout 200 python tools/fir_map_probe.py --configs map1,free,w4_b100 --channels $1 --rows $2 --taps $3 --fused $f 2>> $O/probe_shapes.err | sed "s/^{/{\"channels\": $1, \"rows\": $2, /" | tee -a $O/probe_shapes.log | cut -c1-270
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
for c in w4_b100 w4_b50 w4_b200 w4_b0 w8_b100 wall_b100; do pmc $c --only $c; done
pmc w4_b100_fma --only w4_b100 --fused 1
for shape in "4096 524288 256" "16384 131072 256" "32768 65536 256" "8192 262144 128"; do
  set -- $shape
  pmc w4_b100_c$1_t$3 --only w4_b100 --channels $1 --rows $2 --taps $3
  pmc free_c$1_t$3 --only free --channels $1 --rows $2 --taps $3
done
