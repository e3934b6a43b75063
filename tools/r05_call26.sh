#!/bin/bash
# Round 5, call 26: what FETCH_SIZE reports for a known byte count read in k_fir_ring's shape (tools/ubench_fetch8.hip).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05z
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | tee $O/smi.log
for ctr in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | tr ' ' '+')
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/raw_$tag -o p -- $R/tools/variants/ubench_fetch8 > $O/run_$tag.log 2>&1
  python - $O/raw_$tag <<'P' | tee -a $O/fetch8.log
import collections, csv, glob, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
by_disp = {}
for k, cs in per.items():
  for c, v in cs.items():
    print("%-12s %-24s launches %d  mean per launch %.6g" % (k, c, len(v), sum(v) / len(v)))
P
  rm -rf $O/raw_$tag
done
grep "bytes read" $O/run_FETCH_SIZE.log | tee -a $O/fetch8.log
