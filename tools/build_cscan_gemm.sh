#!/bin/bash
# (EXTRA="-DCGEMM_NS=4 -DCGEMM_SPLIT=8" NAME=_n4s8 for other tilings)
# Variant library with tools/experiments/alz_scan_gemm.hip in place of csrc/alz_scan.hip (the other objects are the shipped
# build's: make -C audiolazy_amd/csrc first) -> tools/variants/libalzhip_cgemm${NAME}.so; use: ALZ_CSCAN_GEMM=1 ALZ_LIBRARY=... python bench.py
#   tools/gpu_call.sh TAG lib:cgemm,ALZ_CSCAN_GEMM=1:--workload+gammatone+--streams+1+--log2-samples+20+--time-parallel+1
R=$(cd $(dirname $0)/.. && pwd)
C=$R/audiolazy_amd/csrc
V=$R/tools/variants
mkdir -p $V
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -DALZ_TUNING $EXTRA -I$C -I$R/include \
  -c $R/tools/experiments/alz_scan_gemm.hip -o $V/alz_scan_gemm${NAME}.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/alz.map $(ls $C/*.o | grep -v "/alz_scan.o") $V/alz_scan_gemm${NAME}.o -ldl \
  -o $V/libalzhip_cgemm${NAME}.so && echo "built cgemm$NAME"
