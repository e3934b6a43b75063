#!/bin/bash
# Round-end measurement pass (round 4): the gpu suite, the driver's command, rocprofv3 kernel stats + dispatch rows of the
# same command, the per-workload PMC traffic table on the shipped binary, and the GPU-side differential fuzzers.
#   gpurun --timeout 2400 -- 'bash tools/r04_final.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
bash tools/gpu_call.sh r04final tests bench stats:--no-cpu-baseline+--no-parity-check
for f in "fuzz_bank.py 250 404" "fuzz_outer.py 120 405" "fuzz_stream.py 150 406"; do
  set -- $f
  timeout 300 python tools/$1 $2 $3 > $O/${1%.py}.log 2>&1; echo "$1 rc=$? $(tail -1 $O/${1%.py}.log | cut -c1-160)"
done
bash tools/pmc_workloads.sh r04final/pmc > $O/pmc_workloads.log 2>&1
python tools/pmc_table.py $O/pmc $R/profiles/r03_pmc_traffic_table.json > $O/r04_pmc_traffic_table.json 2> $O/pmc_table.err
tail -20 $O/pmc_workloads.log | cut -c1-200
