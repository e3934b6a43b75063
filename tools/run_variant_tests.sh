#!/bin/bash
# a selection of the gpu suite through a tools/variants library:  run_variant_tests.sh NAME "-k expr"
cd ${GRAFT_REPO_ROOT:-.}
ALZ_LIBRARY=$PWD/tools/variants/libalzhip_$1.so timeout 500 python -m pytest tests -m gpu -q -x "${@:2}" 2>&1 | tail -3
