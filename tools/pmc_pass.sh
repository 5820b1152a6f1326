#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...> -- <python script args...>
# One rocprofv3 --pmc pass (counters collected with --kernel-trace only) of tools/time_render.py.
set -e
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/time_render.py "$@" > $out.log 2>&1
