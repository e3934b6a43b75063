#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counters...> -- <command...>
# One rocprofv3 counter pass (PMC only + kernel trace, CSV) written under gpurun_out/pmc_<tag>/.
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
