#!/bin/bash
# the shipped library's clocked shapes on a fresh box, three runs each (first runs included: one box in three collapsed the FMA shape to
# the free-running 303 Gsamples/s on its first run at 5750 GB/s): appended to gpurun_out/r06_pace_check/check.log, one section per box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_check; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json -"
{ echo "== box $(rocm-smi --showuniqueid 2>/dev/null | grep 'GPU\[' | head -1 | sed 's/.*: //')"
for rep in 1 2 3; do
for a in "--workload biquad --fused" "--workload envelope" "--workload timevar --streams 0" "--workload biquad --channels 8192 --log2-samples 19" "--workload comb" "--workload biquad"; do
  timeout 300 python bench.py $B $a > $O/l.json 2> $O/l.err || tail -3 $O/l.err
  echo "shipped $a: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-100)"
done
done; } 2>&1 | tee $O/check_$(date +%s).log
