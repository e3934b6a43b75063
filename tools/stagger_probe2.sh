#!/bin/bash
# stagger_start on the cascade kernels (tuning variant of alz_casc.hip, ALZ_CASC_STAGGER ticks per workgroup phase, 64 phases) and
# small values for k_duo (tuning variant of alz_wave.hip, ALZ_DUO_STAGGER)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_stagger2; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-parity-check --full-json -"
{
for rep in 1 2; do
for st in 0 2 5 20 80 320; do
  export ALZ_LIBRARY=$R/tools/variants/libalzhip_casc_tune.so ALZ_CASC_STAGGER=$st
  for m in "--workload gammatone --steps 60 --warmup 30" "--workload gammatone --bank-layout time --steps 60 --warmup 30" "--workload gammatone --streams 1 --log2-samples 20 --time-parallel 1 --bank-layout chan --steps 30 --warmup 10" "--workload gammatone --streams 1 --log2-samples 20 --time-parallel 1 --bank-layout time --steps 30 --warmup 10" "--workload gammatone --streams 256 --steps 10 --warmup 3"; do
    timeout 300 python bench.py $m $B > $O/l.json 2> $O/l.err || tail -3 $O/l.err
    echo "casc stagger $st [$m]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
  done
done
done
unset ALZ_CASC_STAGGER
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
for st in 0 1 2 5 10 20; do
  echo "== ALZ_DUO_STAGGER=$st"
  ALZ_DUO_STAGGER=$st timeout 500 python tools/xy_offset_probe.py chan 2>&1 | grep "^chan " | awk -v st=$st '{ms[$2] = ms[$2] " " $12} END{for (k in ms) print "duo stagger", st, k, "ms by offset:", ms[k]}'
done
} 2>&1 | tee $O/stagger2.log
