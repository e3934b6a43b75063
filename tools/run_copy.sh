#!/bin/bash
# builds (if needed) and runs tools/ubench_copy on the GPU box
cd ${GRAFT_REPO_ROOT:-.}
[ -x tools/ubench_copy ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o tools/ubench_copy
timeout 200 tools/ubench_copy
