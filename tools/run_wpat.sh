#!/bin/bash
# builds (if needed) and runs tools/ubench_wpat on the GPU box
cd ${GRAFT_REPO_ROOT:-.}
[ -x tools/ubench_wpat ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_wpat.hip -o tools/ubench_wpat
timeout 200 tools/ubench_wpat
