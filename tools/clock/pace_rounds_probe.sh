#!/bin/bash
# k_tvpc / k_comb_tm (one workgroup per CU) on the tile clock when the launch is several FULL rounds of workgroups (8192, 12288 channels),
# against the library before (tools/variants/libalzhip_before_rounds.so: such launches ran free); 6144 and 5120 channels must not change
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_pace_rounds; mkdir -p $O
B="--no-cpu-baseline --no-secondary --steps 10 --warmup 3 --full-json -"
for rep in 1 2; do
  for c in 4096 5120 6144 7680 8192 12288; do
    for lib in before_rounds shipped; do
      if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
      timeout 300 python bench.py $B --workload timevar --streams 0 --channels $c --log2-samples 18 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
      echo "timevar per channel $c ch [$lib]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-135)"
      timeout 300 python bench.py $B --workload comb --channels $c --log2-samples 18 > $O/l.json 2> $O/l.err || tail -3 $O/l.err
      echo "comb_fb $c ch [$lib]: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-135)"
    done
  done
done 2>&1 | tee $O/rounds.log
unset ALZ_LIBRARY
timeout 900 python -m pytest tests/test_gpu_bank.py tests/test_gpu_timevar.py -q -m gpu 2>&1 | tail -3
