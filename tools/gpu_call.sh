#!/bin/bash
# One parameterised GPU-box call (replaces round 2's 40 one-off gpu_call_*.sh):
#   gpurun --timeout T -- 'tools/gpu_call.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/.  Steps:
#   tests[:ARGS]     the -m gpu suite (ARGS: extra pytest arguments, e.g. -k+fir)
#   quick:ARGS       a selection of the suite under a 4-minute limit (new kernels: a hang must not eat the call)
#   sh:SCRIPT        bash tools/SCRIPT
#   bench[:ARGS]     python bench.py ARGS           (':' separates, '+' stands for a blank)
#   tune:ENV:ARGS    the same through the -DALZ_TUNING library (tools/variants/libalzhip_tuning.so) with ENV set
#   abl:ENV:ARGS     the same through the -DALZ_ABLATE library (timing ablations: WRONG output by design)
#   lib:NAME[,ENV]:ARGS  python bench.py ARGS through tools/variants/libalzhip_NAME.so (with ENV set)
#   py:SCRIPT[:ENV]  python tools/SCRIPT
#   stats[:ARGS]     rocprofv3 --kernel-trace --stats of python bench.py ARGS -> kernel_stats.csv / kernel_dispatches.csv
#   pmc:CTR:ARGS     one rocprofv3 --pmc CTR pass (+ kernel trace only) of python bench.py ARGS, summarised per kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
i=0
for step in "$@"; do
  i=$((i + 1))
  IFS=':' read -r kind a b <<< "$step"
  a=${a//+/ }; b=${b//+/ }
  cd $R
  case $kind in
    tests) timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 $a > $O/pytest_gpu_$i.log 2>&1; tail -4 $O/pytest_gpu_$i.log ;;
    quick) timeout 240 python -m pytest tests -m gpu -q -x $a > $O/quick_$i.log 2>&1; echo "quick [$a] rc=$?"; tail -4 $O/quick_$i.log ;;
    sh)    timeout 600 bash tools/$a > $O/sh_$i.log 2>&1; echo "sh [$a] rc=$?"; tail -30 $O/sh_$i.log ;;
    bench) timeout 900 python bench.py $a > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench [$a] rc=$?"; python tools/show_line.py $O/bench_$i.json ;;
    tune)  env $a ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so timeout 900 python bench.py $b > $O/tune_$i.json 2> $O/tune_$i.err
           echo "tune [$a] [$b] rc=$?"; python tools/show_line.py $O/tune_$i.json ;;
    abl)   env $a ALZ_LIBRARY=$R/tools/variants/libalzhip_ablate.so timeout 300 python bench.py $b > $O/abl_$i.json 2> $O/abl_$i.err
           echo "abl [$a] [$b] rc=$?"; python tools/show_line.py $O/abl_$i.json ;;
    lib)   IFS=',' read -r nm ev <<< "$a"
           env $ev ALZ_LIBRARY=$R/tools/variants/libalzhip_$nm.so timeout 900 python bench.py $b > $O/lib_$i.json 2> $O/lib_$i.err
           echo "lib [$a] [$b] rc=$?"; python tools/show_line.py $O/lib_$i.json ;;
    py)    env $b timeout 900 python tools/$a > $O/py_$i.log 2>&1; echo "py [$a] rc=$?"; tail -30 $O/py_$i.log ;;
    stats) cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$i -o s -- python $R/bench.py $a > $O/stats_$i.log 2>&1
           find $O/stats_$i -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$i.csv \;
           find $O/stats_$i -name "*kernel_trace.csv" -exec sh -c 'grep -E "Kernel_Name|alz::" "$1" > '$O'/kernel_dispatches_'$i'.csv' _ {} \;
           rm -rf $O/stats_$i; head -14 $O/kernel_stats_$i.csv | cut -c1-170 ;;
    pmc)   cd /tmp; timeout 900 rocprofv3 --pmc $a --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python $R/bench.py $b > $O/pmc_$i.log 2>&1
           python $R/tools/pmc_summary.py $O/pmc_$i > $O/pmc_${i}_${a// /_}.txt 2>&1; rm -rf $O/pmc_$i; tail -25 $O/pmc_${i}_${a// /_}.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
