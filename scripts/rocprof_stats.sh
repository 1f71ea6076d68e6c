#!/bin/bash
# usage (on the GPU box): scripts/rocprof_stats.sh <tag> <workload> [steps] -> gpurun_out/<tag>/rocprof_<workload>_kernel_stats.csv
# rocprofv3 --kernel-trace --stats of one bench.py workload (program directly after `--`).
tag=$1; wl=$2; steps=${3:-50}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/rp_$wl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_$wl -o p -- python3 $root/bench.py --no-cpu-baseline --no-configs --workload $wl --steps $steps --warmup 5 > $out/rp_$wl.log 2>&1
f=$(find $out/rp_$wl -name "p_kernel_stats.csv" | head -1)
cp "$f" $out/rocprof_${wl}_kernel_stats.csv
rm -rf $out/rp_$wl
head -30 $out/rocprof_${wl}_kernel_stats.csv | cut -c1-150
