#!/bin/bash
# k_cdot3 with band groups pinned to XCDs (ALZ_CDOT_XCD=1, shipped) against the launch's natural order (0): the one-stream filterbank in
# time-parallel mode, both layouts, interleaved; then the per-kernel times of both under rocprofv3
cd ${GRAFT_REPO_ROOT:-$(pwd)}
R=$(pwd); O=$R/gpurun_out/r06_sh; mkdir -p $O
A="--workload gammatone --streams 1 --time-parallel 1 --log2-samples 20 --no-cpu-baseline --no-secondary --steps 40 --warmup 10 --full-json -"
for rep in 1 2; do
  for v in 1 0; do
    for lay in chan time; do
      ALZ_CDOT_XCD=$v ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so python bench.py $A --bank-layout $lay > /tmp/c.json 2> /tmp/c.err || tail -3 /tmp/c.err
      echo "xcd_map=$v $lay: $(python tools/show_line.py /tmp/c.json | head -1 | cut -c1-100)"
    done
  done
done
export TMPDIR=/tmp
for v in 1 0; do
  (cd /tmp; ALZ_CDOT_XCD=$v ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cd_$v -o s -- python $R/bench.py $A --no-parity-check > $O/cd_$v.log 2>&1)
  echo "xcd_map=$v kernel stats:"; find $O/cd_$v -name "*kernel_stats.csv" -exec grep -E "k_cdot3|k_casc|k_cscan_fix" {} \; | cut -c1-150; rm -rf $O/cd_$v
done
