#!/bin/bash
# builds (if needed) and runs tools/ubench_clock on the GPU box
cd ${GRAFT_REPO_ROOT:-.}
[ -x tools/ubench_clock ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_clock.hip -o tools/ubench_clock
timeout 120 tools/ubench_clock
