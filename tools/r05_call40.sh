#!/bin/bash
# Round 5, call 40: the L2's own hit / miss counters (128-byte lines, tools/ubench_fetch8.hip) for k_fir_ring on configs[2]:
# interleaved runs against the shipped chains, both arithmetic modes (-DALZ_TUNING build for the A/B).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05an
mkdir -p $O
cd $R
export TMPDIR=/tmp
export ALZ_LIBRARY=$R/tools/variants/libalzhip_tuning.so
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
for cfg in map1 auto free; do
  for f in 0 1; do
    cd /tmp
    timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/raw -o p -- python $R/tools/fir_map_probe.py --only $cfg --fused $f > $O/raw.log 2>&1
    python - $O/raw $cfg $f <<'P' | tee -a $O/l2_hit_miss.log
import collections, csv, glob, sys
per = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(fn)):
    if "k_fir_ring" in r["Kernel_Name"]:
      per[r["Counter_Name"]].append(float(r["Counter_Value"]))
h, m = (sum(per[k]) / max(len(per[k]), 1) for k in ("TCC_HIT_sum", "TCC_MISS_sum"))
print("%-5s fused=%s  launches %d  lines per launch: hit %.4g  miss %.4g  hit rate %.3f  missed bytes %.1f GB (input 17.18)" % (
    sys.argv[2], sys.argv[3], len(per["TCC_HIT_sum"]), h, m, h / (h + m) if h + m else 0, m * 128 / 1e9))
P
    rm -rf $O/raw
  done
done
