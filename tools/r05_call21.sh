#!/bin/bash
# Round 5, call 21: what FETCH_SIZE counts for k_fir_ring's 8-byte-per-lane buffer loads (1 tap: every input byte read
# exactly once), how it grows with the tap count, and whether the 64 KiB (power of two) row pitch is why neighbouring
# runs do not find each other's rows in L2 (call 20: 2.81 x the input whatever the run-to-wave mapping).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05u
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
timeout 200 python tools/fir_map_probe.py --pad 16 > $O/probe_pad16.log 2> $O/probe_pad16.err; echo "probe rc=$?"; cut -c1-250 $O/probe_pad16.log
pmc() {  # key, probe args
  key=$1; shift
  cd /tmp
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/raw_$key -o p -- python $R/tools/fir_map_probe.py "$@" > $O/raw_$key.log 2>&1
  python $R/tools/pmc_sum.py $O/raw_$key FETCH_SIZE 3 > $O/pmc_$key.json 2>> $O/errors.log
  rm -rf $O/raw_$key
  echo "$key: $(cut -c1-200 $O/pmc_$key.json)"
}
pmc taps1 --only map1 --taps 1
pmc taps16 --only map1 --taps 16
pmc taps64 --only map1 --taps 64
pmc taps128 --only map1 --taps 128
pmc taps256_pad16 --only map1 --pad 16
pmc taps256_pad16_cohort_stag --only cohort_stag --pad 16
pmc taps256_pad16_cohort_stag_nty --only cohort_stag_nty --pad 16
pmc taps256_pad264 --only map1 --pad 264
pmc taps256_c2048 --only map1 --channels 2048 --rows 1048576
