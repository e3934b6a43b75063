#!/bin/bash
# Round-end measurement pass on the GPU box:
#   1. the whole gpu test suite;
#   2. the driver's command (python bench.py): primary line + secondaries + CPU legs;
#   3. rocprofv3 --kernel-trace --stats of the same command (kernel stats + full dispatch rows);
#   4. the two PMC passes behind roofline.traffic at the BENCHMARKED block size (2^20 samples);
#   5. side measurements (channel-major layout, formats / time-varying kernels, host path).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_full.json 2>$O/bench_full.err
echo "bench rc=$?"
timeout 200 python bench.py --layout chan --no-cpu-baseline --no-secondary > $O/bench_chan.json 2>/dev/null
timeout 200 python bench.py --fused --no-cpu-baseline --no-secondary > $O/bench_fused.json 2>/dev/null
timeout 200 python bench.py --channels 512 --time-parallel 1 --no-cpu-baseline --no-secondary > $O/bench_narrow_tp.json 2>/dev/null
timeout 200 python tools/io_time.py > $O/io_time.log 2>&1
timeout 200 python tools/tv_time.py > $O/tv_time.log 2>&1
timeout 200 python tools/host_path_time.py > $O/host_path.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline --no-parity-check > $O/stats.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-parity-check > $O/pmc_$ctr.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE k_duo 4096 1048576 > $O/pmc_traffic.json 2>$O/pmc_traffic.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# dispatch rows of this library's kernels with EVERY column (LDS size, workgroup / grid size, registers)
find $O/stats -name "*kernel_trace.csv" -exec sh -c 'grep -E "Kernel_Name|alz::" "$1" > '$O'/kernel_dispatches.csv' _ {} \;
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench_full.json'))
print(round(d['value'],1), round(d['roofline']['frac'],4), d['config']['parity_spot_check'])
for k,v in d.get('secondary',{}).items(): print(' ', k, round(v['value'],3), v['unit'], round(v['roofline']['frac'],3), v['kernel'][:40], '|', v['parity'][:70])
PY
cat $O/pmc_traffic.json | head -20
head -12 $O/kernel_stats.csv | cut -c1-160
