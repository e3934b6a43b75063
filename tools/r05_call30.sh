#!/bin/bash
# Round 5, call 30: chain width of k_fir_ring's paced chains (bounded waits) against bank width and tap count; the
# constants around the best point of call 29 (W = 8 on configs[2]).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ad
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
for f in 0 1; do
  timeout 240 python tools/fir_map_probe.py --fused $f --configs map1,free,w4,w8,w16,w8_b50,w8_b150,w8_s70,w8_s100,w8_h0,w8_cfill8,w8_cfill32 2>> $O/probe.err | tee -a $O/probe_8192.log | cut -c1-250
done
for shape in "2048 1048576 256" "4096 524288 256" "16384 131072 256" "32768 65536 256" "8192 262144 128" "8192 262144 512" "8192 262144 96" "8192 98304 256"; do
  set -- $shape
  for f in 0 1; do
    timeout 200 python tools/fir_map_probe.py --configs map1,free,w2,w4,w8,w16 --channels $1 --rows $2 --taps $3 --fused $f 2>> $O/probe_shapes.err | sed "s/^{/{\"channels\": $1, \"rows\": $2, /" | tee -a $O/probe_shapes.log | cut -c1-270
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
pmc w8 --only w8
pmc w8_fma --only w8 --fused 1
pmc w8_b50 --only w8_b50
pmc w8_b150 --only w8_b150
for shape in "2048 1048576 256" "4096 524288 256" "16384 131072 256"; do
  set -- $shape
  pmc w8_c$1_t$3 --only w8 --channels $1 --rows $2 --taps $3
done
pmc w4_c16384_t256 --only w4 --channels 16384 --rows 131072
pmc w2_c32768_t256 --only w2 --channels 32768 --rows 65536
pmc w16_c8192_t512 --only w16 --taps 512
pmc w8_c8192_t512 --only w8 --taps 512
pmc w4_c8192_t128 --only w4 --taps 128
pmc w8_c8192_t128 --only w8 --taps 128
