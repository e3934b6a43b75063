#!/bin/bash
# one-pole banks with and without AUX pacing (tuning library: ALZ_DUO_AUXPACE), over block lengths, widths, layouts
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/tools/variants/libalzhip_tuning.so
for a in "20 4096 time pole (" "20 4096 chan pole (" "18 4096 time pole (" "16 4096 time pole (" "20 5120 time pole ("; do
  for pace in ${PACES:-0 2 3 17}; do
    echo "pace $pace: $(ALZ_DUO_AUXPACE=$pace ALZ_LIBRARY=$L python tools/pattern_sweep.py $a 2>/dev/null | tr '\n' ' ' | sed 's/Gsamples\/s/|/g' | cut -c1-230)"
  done
done
