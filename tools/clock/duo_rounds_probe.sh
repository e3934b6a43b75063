#!/bin/bash
# k_duo on the tile clock in rounds of 512 workgroups (16 384, 24 576, 32 768 channels; ALZ_DUO_ROUNDS_MAX=4) against k_wave<64> (=1, the
# shipped dispatch), bit-exact and FMA; tools/variants/libalzhip_wave_tune.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_duo_rounds; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
for rep in 1 2; do
  for cn in "16384 18" "24576 17" "32768 17"; do
    set -- $cn
    for m in "" "--fused"; do
      for r in 1 4; do
        ALZ_G=$([ $r = 4 ] && echo 16 || echo 0) ALZ_DUO_ROUNDS_MAX=$r timeout 300 python bench.py $B --workload biquad $m --channels $1 --log2-samples $2 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
        echo "biquad $1 ch x 2^$2 $m [rounds max $r]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-120)"
      done
    done
  done
done 2>&1 | tee $O/duo_rounds.log
