#!/bin/bash
# the FMA mode on channel-major banks of several widths: two-wave FMA kernel (LIB unset / "old" library) against the FMA kernel with
# the storing wave and non-temporal tiles
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_duofmaw; mkdir -p $O
for rep in 1 2; do
  for w in ${WIDTHS:-512 1024 2048 4096 5120 6144}; do
    for lib in ${LIBS:-shipped duo_fma3}; do
      if [ $lib = shipped ]; then unset ALZ_LIBRARY; else export ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so; fi
      timeout 300 python bench.py --workload biquad --fused --layout chan --channels $w --log2-samples ${LOG2:-20} --no-cpu-baseline --no-secondary --no-parity-check --steps 10 --warmup 3 --full-json - > $O/l.json 2> $O/l.err || tail -3 $O/l.err
      echo "$lib $w: $(python tools/show_line.py $O/l.json | head -1 | cut -c1-110)"
    done
  done
done 2>&1 | tee $O/duo_fma_cm_widths.log
