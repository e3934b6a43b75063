#!/bin/bash
# Round 5, GPU call 6: the broadcast replay with a two-slot output ring (two one-wave workgroups per SIMD fit) -- chunk sweep again.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | head -1 | tee $O/smi.log
timeout 300 python -m pytest -m gpu -q -x --timeout=300 tests/test_gpu_cscan_dot.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 300 python tools/tp_chunk_sweep.py > $O/chunk_sweep.log 2>&1; cat $O/chunk_sweep.log | cut -c1-160
