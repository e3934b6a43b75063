#!/bin/bash
# does the headline figure depend on how long the chip has been busy (the clock ramp that k_pipe showed, profiles/r06_pipe_long.log)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_headline_pace; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-parity-check --full-json -"
for rep in 1 2 3; do
  for sw in "20 5" "20 50" "20 200" "100 5"; do set -- $sw
    timeout 300 python bench.py $B --steps $1 --warmup $2 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
    echo "headline [steps $1 warmup $2]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-90)"
  done
done 2>&1 | tee $O/headline_warm.log
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
