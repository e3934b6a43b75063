#!/bin/bash
# Round 5, call 39: the four GPU-side differential fuzzers on the round's last library (the FIR mapping and the FMA mode's
# dispatch changed after the final pass that ran them).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05am
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
for f in "fuzz_timeparallel.py 100 1404" "fuzz_bank.py 200 1405" "fuzz_outer.py 100 1406" "fuzz_stream.py 120 1407"; do
  set -- $f
  timeout 150 python tools/$1 $2 $3 > $O/${1%.py}.log 2>&1; echo "$1 rc=$? $(tail -1 $O/${1%.py}.log | cut -c1-200)"
done
