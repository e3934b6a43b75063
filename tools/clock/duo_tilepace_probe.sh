#!/bin/bash
# k_duo, time-major, FMA mode on a chip-wide streaming block: the three-wave fused kernel with every workgroup's tile requests on one
# clock (WArgs::tile_pace; launch_wave_impl).  Through the tuning build of alz_wave.hip (tools/build_variant.sh duo_tune alz_wave.hip
# -DALZ_TUNING): ALZ_DUO_PACED_FMA=0 is the rule before (default kernel in the FMA mode for this shape), ALZ_DUO_PACE_GBPS the rate
# the clock is derived from (0: free-running), ALZ_DUO_TILEPACE an absolute pace in 1/16 ticks of 10 ns.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_tilepace; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_duo_tune.so
B="--workload biquad --fused --no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json -"
one() { # label, env, extra args
  env $2 timeout 300 python bench.py $B $3 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "$1 [$2] $3: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
}
for rep in 1 2; do
  one "rule before" ALZ_DUO_PACED_FMA=0 ""
  for g in ${RATES:-0 5500 5650 5700 5750 5800 5850 5900 6000}; do one "4096 ch" ALZ_DUO_PACE_GBPS=$g ""; done
  one "rule before" ALZ_DUO_PACED_FMA=0 "--channels 5120"
  for g in ${RATES5:-0 5000 5300 5500 5750}; do one "5120 ch" ALZ_DUO_PACE_GBPS=$g "--channels 5120"; done
  one "rule before" ALZ_DUO_PACED_FMA=0 "--log2-samples 18"
  for g in 0 5750; do one "4096 ch 2^18" ALZ_DUO_PACE_GBPS=$g "--log2-samples 18"; done
  one "rule before" ALZ_DUO_PACED_FMA=0 "--log2-samples 14"
  for g in 0 5750; do one "4096 ch 2^14" ALZ_DUO_PACE_GBPS=$g "--log2-samples 14"; done
done 2>&1 | tee $O/tilepace_shipped_form.log
unset ALZ_LIBRARY
timeout 900 python -m pytest tests/test_gpu_fullwidth.py -x -q -m gpu -k "fused or cfg2" 2>&1 | tail -5 | tee $O/tests.log
timeout 300 python bench.py --workload biquad --fused --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json - 2> $O/s.err | python tools/show_line.py /dev/stdin | head -2 | tee -a $O/tests.log
