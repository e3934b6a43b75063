#!/bin/bash
# Round 5, call 31: launch_fir's rule for k_fir_ring's chains (width 8 / 4 / unpaced by bank width) against the
# interleaved mapping over bank widths, tap counts and block lengths (-DALZ_TUNING build for the A/B), the wait bound in
# the FMA mode, then the FIR tests on the SHIPPED library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ae
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 900 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_bank.py -x -q -k "fir" > $O/pytest_fir.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_fir.log
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
for f in 0 1; do
  timeout 240 python tools/fir_map_probe.py --fused $f --configs map1,auto,free,b50,b150,b200,b300,h100,s100 2>> $O/probe.err | tee -a $O/probe_8192.log | cut -c1-250
done
for shape in "512 4194304 256" "1024 2097152 256" "2048 1048576 256" "4096 524288 256" "16384 131072 256" "32768 65536 256" "8192 262144 96" "8192 262144 128" "8192 262144 512" "8192 98304 256" "8192 49152 256"; do
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
pmc auto --only auto
for c in auto b150 b200 b300 h100; do pmc fma_$c --only $c --fused 1; done
for shape in "1024 2097152 256" "2048 1048576 256"; do
  set -- $shape
  pmc map1_c$1 --only map1 --channels $1 --rows $2 --taps $3
  pmc auto_c$1 --only auto --channels $1 --rows $2 --taps $3
done
