#!/bin/bash
# Round-end measurement pass (round 5): the gpu suite, the driver's command, rocprofv3 kernel stats + dispatch rows of the
# same command, the per-workload PMC traffic table on the shipped binary, and the GPU-side differential fuzzers.
#   gpurun --timeout 2400 -- 'bash tools/r05_final.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05final
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | head -1 | tee $O/smi.log
bash tools/gpu_call.sh r05final tests bench:--gpus+1+--steps+20+--warmup+5 stats:--no-cpu-baseline+--no-parity-check
cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
for f in "fuzz_timeparallel.py 100 1404" "fuzz_bank.py 200 1405" "fuzz_outer.py 100 1406" "fuzz_stream.py 120 1407"; do
  set -- $f
  timeout 400 python tools/$1 $2 $3 > $O/${1%.py}.log 2>&1; echo "$1 rc=$? $(tail -1 $O/${1%.py}.log | cut -c1-200)"
done
bash tools/pmc_workloads.sh r05final/pmc > $O/pmc_workloads.log 2>&1
python tools/pmc_table.py $O/pmc $R/profiles/r04_pmc_traffic_table.json > $O/r05_pmc_traffic_table.json 2> $O/pmc_table.err
tail -22 $O/pmc_workloads.log | cut -c1-200
