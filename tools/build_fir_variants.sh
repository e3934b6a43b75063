#!/bin/bash
# A/B builds of the shared-tap FIR kernel's register tile (same ABI; loaded through ALZ_LIBRARY):
#   tools/variants/fir_R<outputs per lane>_K<taps per block>_W<waves per SIMD>.so
cd "$(dirname "$0")/../audiolazy_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math"
OTHERS=$(ls *.o | grep -v alz_fir.o)
for cfg in "16 8 3" "16 8 4" "32 8 3" "24 8 2" "24 8 3" "48 8 2" "32 16 2" "64 8 2"; do
  set -- $cfg
  out=../../tools/variants/fir_R$1_K$2_W$3.so
  /opt/rocm/bin/hipcc $FLAGS -DALZ_FIR_R=$1 -DALZ_FIR_SK=$2 -DALZ_FIR_K=16 -DALZ_FIR_WAVES=$3 -c alz_fir.hip -o /tmp/fir_var.o 2>/tmp/fir_var.err \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/fir_var.o $OTHERS -o $out && echo "built $out" || { echo "FAILED $cfg"; tail -3 /tmp/fir_var.err; }
done
