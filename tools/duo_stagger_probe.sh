#!/bin/bash
# channel-major k_duo: workgroups started a few tiles apart (ALZ_DUO_STAGGER ticks of 10 ns per workgroup index, tuning variant of
# alz_wave.hip) against the lock-step start, over the x / y placements of tools/xy_offset_probe.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_stagger; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
for st in ${STAGGERS:-0 20 80 160 400}; do
  echo "== ALZ_DUO_STAGGER=$st"
  ALZ_DUO_STAGGER=$st timeout 500 python tools/xy_offset_probe.py chan 2>&1 | grep "^chan " | awk -v st=$st '{ms[$2] = ms[$2] " " $12} END{for (k in ms) print "stagger", st, k, "ms by offset:", ms[k]}'
done 2>&1 | tee $O/duo_stagger.log
