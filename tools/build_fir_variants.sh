#!/bin/bash
# A/B builds of the FIR ring kernel's register tile / prefetch depth into tools/variants/ (loaded with ALZ_LIBRARY=):
#   usage: tools/build_fir_variants.sh   (needs `make -C audiolazy_amd/csrc tuning` first: the other objects are reused)
set -e
cd $(dirname $0)/../audiolazy_amd/csrc
V=../../tools/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -DALZ_TUNING"
build() {  # name, defines
  mkdir -p $V/obj_$1
  cp $V/obj_tuning/*.o $V/obj_$1/
  /opt/rocm/bin/hipcc $FLAGS $2 -c alz_fir.hip -o $V/obj_$1/alz_fir.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=alz.map $V/obj_$1/*.o -o $V/libalzhip_$1.so
  rm -rf $V/obj_$1
}
build fir_k8pf1 "-DALZ_FIR_RING_K=8 -DALZ_FIR_RING_PF=1" &
build fir_k8pf2r40 "-DALZ_FIR_RING_K=8 -DALZ_FIR_RING_PF=2 -DALZ_FIR_RING_R=40" &
build fir_k4pf4 "-DALZ_FIR_RING_K=4 -DALZ_FIR_RING_PF=4" &
build fir_k4pf2 "-DALZ_FIR_RING_K=4 -DALZ_FIR_RING_PF=2" &
wait
ls -la $V/*.so
