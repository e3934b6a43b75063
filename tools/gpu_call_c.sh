#!/bin/bash
# round-2 GPU call C: new tests first, then the whole suite, the bench line, PMC passes on the FMA FIR kernel
mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_maps.py tests/test_gpu_fullwidth.py -q > gpurun_out/r02c/pytest_new.log 2>&1
tail -25 gpurun_out/r02c/pytest_new.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_scan.py --deselect tests/test_gpu_maps.py --deselect tests/test_gpu_fullwidth.py > gpurun_out/r02c/pytest_rest.log 2>&1
tail -8 gpurun_out/r02c/pytest_rest.log
timeout 600 python bench.py > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err
echo "bench rc=$?"; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r02c/bench.json'))
print(d['value'], d['roofline']['frac'], d['config']['parity_spot_check'])
for k,v in d.get('secondary',{}).items(): print(k, round(v['value'],3), v['unit'], round(v['roofline']['frac'],3), v['kernel'], '|', v['parity'][:90])
print({k:round(v['value'],5) for k,v in d['cpu_baseline']['legs'].items()}, d['cpu_baseline'].get('usable_cores'))
PY
cd /tmp
for ctr in "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/r02c/pmc_firfma_$tag -o p -- python $R/bench.py --workload fir --fused --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check > $R/gpurun_out/r02c/pmc_firfma_$tag.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/r02c/pmc_firfma_GRBM_GUI_ACTIVE gpurun_out/r02c/pmc_firfma_SQ_INSTS_VALU 2>&1 | grep -v "^$" | head -40
