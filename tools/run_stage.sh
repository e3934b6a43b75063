#!/bin/bash
# runs tools/ubench_stage_ilv (built in the container: it includes audiolazy_amd/csrc/alz_casc.hip) on the GPU box
cd ${GRAFT_REPO_ROOT:-.}
timeout 120 tools/ubench_stage_ilv
