#!/bin/bash
# usage: scripts/pmc.sh <outname> <counters...> -- <bench args>
# collects PMC counters in their own rocprofv3 pass (no trace domains besides kernel dispatch)
name=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/$name.err
ls $GRAFT_REPO_ROOT/gpurun_out/$name/*/ 2>/dev/null | head
