#!/bin/bash
# A variant library that differs from the shipped one in ONE source file's -D flags:
#   tools/build_variant.sh NAME alz_lpc.hip "-DALZ_LPC_PIPE=0"   ->  tools/variants/libalzhip_NAME.so
# (the other objects are the shipped build's: make -C audiolazy_amd/csrc first).  Loaded with ALZ_LIBRARY= by
# tools/gpu_call.sh lib:NAME:ARGS.
R=$(cd $(dirname $0)/.. && pwd)
C=$R/audiolazy_amd/csrc
V=$R/tools/variants
mkdir -p $V
name=$1; src=$2; flags=$3
obj=$V/${src%.hip}_$name.o
cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math $flags -c $src -o $obj 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=alz.map $(ls $C/*.o | grep -v "/${src%.hip}.o") $obj -ldl -o $V/libalzhip_$name.so && echo "built $name"
