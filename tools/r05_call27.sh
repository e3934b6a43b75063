#!/bin/bash
# Round 5, call 27: k_fir_ring's paced chains in their product form (launch_fir picks them; -DALZ_TUNING build so that the
# round-4 mapping and the pacing constants can be set per launch): time against the interleaved mapping, bitwise equality,
# wait statistics and FETCH_SIZE -- at configs[2] and at other bank widths (W = 32 / 8 / 4 waves per chain) and tap counts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05aa
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 240 python tools/fir_map_probe.py > $O/probe_exact.log 2> $O/probe_exact.err; echo "probe rc=$?"; cut -c1-250 $O/probe_exact.log
timeout 200 python tools/fir_map_probe.py --fused 1 > $O/probe_fma.log 2> $O/probe_fma.err; echo "probe fma rc=$?"; cut -c1-250 $O/probe_fma.log
for shape in "4096 524288 256" "16384 131072 256" "32768 65536 256" "8192 262144 128" "8192 262144 64" "8192 65536 256"; do
  set -- $shape
  for f in 0 1; do
    timeout 200 python tools/fir_map_probe.py --configs map1,auto,free --channels $1 --rows $2 --taps $3 --fused $f 2>> $O/probe_shapes.err | sed "s/^{/{\"channels\": $1, \"rows\": $2, /" | tee -a $O/probe_shapes.log | cut -c1-270
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
for c in auto free share70 share100 share100_h150 share85_h150 share85_h400 cfill8 cfill32; do pmc $c --only $c; done
pmc auto_fma --only auto --fused 1
for shape in "4096 524288 256" "16384 131072 256" "32768 65536 256" "8192 262144 128" "8192 262144 64"; do
  set -- $shape
  pmc map1_c$1_t$3 --only map1 --channels $1 --rows $2 --taps $3
  pmc auto_c$1_t$3 --only auto --channels $1 --rows $2 --taps $3
done
