#!/bin/bash
# tools/pattern_sweep.py through each variant library given (names under tools/variants), plus the shipped one
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== shipped"; python tools/pattern_sweep.py 20 2>/dev/null
for v in "$@"; do echo "== $v"; ALZ_LIBRARY=$R/tools/variants/libalzhip_$v.so python tools/pattern_sweep.py 20 2>/dev/null; done
