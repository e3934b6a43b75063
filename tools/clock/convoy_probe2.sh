#!/bin/bash
# second convoy pass: longer checkpoint intervals; the clock AND the convoy (ALZ_CONVOY_CLOCK=1) at the shipped rate and at rates the
# memory system does not follow (does the convoy hold the order where the clock alone collapses?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_convoy; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 120 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
W1="--workload biquad --fused"; W2="--workload envelope"; W3="--workload biquad --channels 8192 --log2-samples 19"
for rep in 1 2; do
  for cv in 528 288 544 320 576 4128; do   # Q 16 S 2; Q 32 S 1; Q 32 S 2; Q 64 S 1; Q 64 S 2; Q 32 S 16
    one "biquad fma" ALZ_CONVOY=$cv "$W1"; one "envelope" ALZ_CONVOY=$cv "$W2"; one "biquad 8192 ch" ALZ_CONVOY=$cv "$W3"
  done
  for cv in 528 544; do
    for g in 0 6000 6400 7000; do   # 0: the shipped rate of each shape
      e="ALZ_CONVOY=$cv,ALZ_CONVOY_CLOCK=1"; [ $g != 0 ] && e="$e,ALZ_DUO_PACE_GBPS=$g"
      one "biquad fma" $e "$W1"; one "envelope" $e "$W2"; one "biquad 8192 ch" $e "$W3"
    done
  done
done 2>&1 | tee $O/convoy2.log
