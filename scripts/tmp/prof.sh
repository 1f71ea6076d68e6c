root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_now
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in hex100c cavity215c; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_$wl -o p -- python3 $root/bench.py --no-cpu-baseline --no-configs --workload $wl --steps 50 --warmup 5 > /dev/null 2>&1
  cp $out/rocprof_$wl/p_kernel_stats.csv $out/rocprof_${wl}_kernel_stats.csv
  rm -rf $out/rocprof_$wl
done
