#!/bin/bash
# is it the stagger or the process?  every value three times, each in a process of its own, FMA mode, three placements
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_stagger3; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so XY_FMA_ONLY=1 XY_OFFS=0,8192,3145728
for rep in 1 2 3; do
  for st in 0 2 5 10 20 40; do
    ALZ_DUO_STAGGER=$st timeout 300 python tools/xy_offset_probe.py chan 2>&1 | grep "^chan " | awk -v st=$st -v rep=$rep '{ms = ms " " $12} END{print "rep", rep, "stagger", st, "fma ms by offset:", ms}'
  done
done 2>&1 | tee $O/stagger3.log
