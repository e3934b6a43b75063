#!/bin/bash
# third convoy pass: the checkpoint in the STORING wave (cfg bit 16: + 65536), which has the time, instead of the wave that requests the tiles
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_convoy; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, args
  env $(echo $2 | tr "," " ") timeout 120 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
W1="--workload biquad --fused"; W2="--workload envelope"; W3="--workload biquad --channels 8192 --log2-samples 19"; W4="--workload biquad --channels 6144 --log2-samples 19"
for rep in 1 2; do
  one "biquad fma" ALZ_CONVOY=0 "$W1"; one "envelope" ALZ_CONVOY=0 "$W2"; one "biquad 8192 ch" ALZ_CONVOY=0 "$W3"; one "biquad 6144 ch" ALZ_CONVOY=0 "$W4"
  for qs in "4 1" "4 2" "8 1" "8 2" "8 4" "16 1" "16 2" "16 4" "32 2"; do
    set -- $qs; cv=$(( 65536 + $1 + $2 * 256 ))
    one "biquad fma Q $1 S $2" ALZ_CONVOY=$cv "$W1"; one "envelope Q $1 S $2" ALZ_CONVOY=$cv "$W2"; one "biquad 8192 ch Q $1 S $2" ALZ_CONVOY=$cv "$W3"; one "biquad 6144 ch Q $1 S $2" ALZ_CONVOY=$cv "$W4"
  done
done 2>&1 | tee $O/convoy3.log
