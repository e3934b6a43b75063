#!/bin/bash
# tools/pattern_sweep.py through the shipped library and each variant library named (tools/variants/libalzhip_NAME.so):
#   PATTERN_ARGS="20 4096 time" bash tools/pattern_lib.sh NAME ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
A=${PATTERN_ARGS:-20}
echo "== shipped"; python tools/pattern_sweep.py $A 2>/dev/null
for v in "$@"; do echo "== $v"; ALZ_LIBRARY=$R/tools/variants/libalzhip_$v.so python tools/pattern_sweep.py $A 2>/dev/null; done
