#!/bin/bash
# Round 5, GPU call 11: channel-major tile DMA with the odd channel's pieces rotated by 128 bytes (ALZ_CM_ROT, variant
# library) -- parity of every channel-major path through the variant, then A/B: the headline bank in [C, N], the narrow
# bank's one-pass form in [C, N], LDS bank conflicts of k_look<CM>.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
export TMPDIR=/tmp
V=$R/tools/variants/libalzhip_cmrot.so
ALZ_LIBRARY=$V timeout 900 python -m pytest -m gpu -q -x --timeout=300 tests/test_gpu_bank.py tests/test_gpu_scan.py tests/test_gpu_fullwidth.py tests/test_gpu_maps.py tests/test_gpu_outer_narrow.py > $O/pytest_rot.log 2>&1; echo "pytest (rotated) rc=$?"; grep -n "passed\|failed" $O/pytest_rot.log | tail -2
for rep in 1 2; do
  for lib in "" cmrot; do
    env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py --layout chan --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > $O/head_${lib:-ship}_$rep.json 2> /dev/null
    echo "headline [C, N] [${lib:-ship}]: $(python tools/show_line.py $O/head_${lib:-ship}_$rep.json | head -1 | cut -c1-120)"
    env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py --channels 512 --time-parallel 1 --layout chan --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > $O/narrow_${lib:-ship}_$rep.json 2> /dev/null
    echo "narrow512 one-pass [C, N] [${lib:-ship}]: $(python tools/show_line.py $O/narrow_${lib:-ship}_$rep.json | head -1 | cut -c1-120)"
    env ${lib:+ALZ_LIBRARY=$R/tools/variants/libalzhip_$lib.so} timeout 300 python bench.py --channels 512 --layout chan --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > $O/narrowbx_${lib:-ship}_$rep.json 2> /dev/null
    echo "narrow512 bit-exact [C, N] [${lib:-ship}]: $(python tools/show_line.py $O/narrowbx_${lib:-ship}_$rep.json | head -1 | cut -c1-120)"
  done
done
ALZ_LIBRARY=$V bash tools/gpu_call.sh r05k "pmc:SQ_LDS_BANK_CONFLICT+SQ_LDS_IDX_ACTIVE+SQ_INSTS_LDS+SQ_WAIT_INST_LDS+SQ_WAVE_CYCLES:--channels+512+--time-parallel+1+--layout+chan+--no-secondary+--no-cpu-baseline+--no-parity-check+--steps+6+--warmup+2" 2>&1 | grep -A6 "k_look" | cut -c1-150 | head -8
