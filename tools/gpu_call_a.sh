#!/bin/bash
# round-2 GPU call A: the whole gpu suite, then the default bench line (with secondaries + CPU legs)
mkdir -p gpurun_out/r02a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -15 gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
echo "bench rc=$?"
tail -c 6000 gpurun_out/r02a/bench.json
tail -5 gpurun_out/r02a/bench.err
