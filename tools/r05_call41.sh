#!/bin/bash
# Round 5, call 41: chains on banks whose group count is not a power of two (the chip then holds a whole number of chains of
# no group: launch_fir's gate keeps the interleaved mapping there) -- forced through the -DALZ_TUNING build, against it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ao
mkdir -p $O
cd $R
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
for shape in "1536 1398080" "4608 466048" "6144 349504" "10240 209664" "12288 174720" "20480 104832"; do
  set -- $shape
  for f in 0 1; do
    timeout 200 python tools/fir_map_probe.py --configs map1,auto,forced,free --channels $1 --rows $2 --fused $f 2>> $O/probe.err | sed "s/^{/{\"channels\": $1, \"rows\": $2, /" | tee -a $O/probe_shapes.log | cut -c1-270
  done
done
