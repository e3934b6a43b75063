#!/bin/bash
# every tap pattern with AUX's pass paced (variant library built with -DALZ_TUNING -DALZ_PACE_ALL=1: tools/build_variant.sh paceall alz_wave.hip ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/tools/variants/libalzhip_paceall.so
for a in "20 4096 time z" "20 4096 chan z"; do
  for pace in ${PACES:-0 1 2 3}; do
    echo "pace $pace: $(ALZ_DUO_AUXPACE=$pace ALZ_LIBRARY=$L python tools/pattern_sweep.py $a 2>/dev/null | tr '\n' ' ' | sed 's/Gsamples\/s/|/g' | cut -c1-260)"
  done
done
