#!/bin/bash
# the headline kernel (two-pole, bit-exact, 4096 channels x 2^20) varies 313 - 330 Gsamples/s between PROCESSES on one box: is the slow
# mode the memory order (then a clock at the fast mode's rate pins it) or the recurrence wave's issue rate (then not)?  Ten processes
# each, interleaved: free-running, 5250 GB/s (13.09 ms), 5350 (12.85 ms); tools/variants/libalzhip_wave_tune.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_headline_pace; mkdir -p $O
export ALZ_LIBRARY=$R/tools/variants/libalzhip_wave_tune.so
B="--no-cpu-baseline --no-secondary --no-parity-check --steps 20 --warmup 5 --full-json -"
for rep in 1 2 3 4 5 6 7 8 9 10; do
  for g in 0 5250 5350; do
    ALZ_DUO_PACE_GBPS=$g timeout 300 python bench.py $B > $O/l.json 2> $O/l.err || tail -3 $O/l.err
    echo "headline [clock $g]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-90)"
  done
done 2>&1 | tee $O/headline_pace.log
