#!/bin/bash
# run on the GPU box: benchmark lines + rocprofv3 kernel statistics + PMC traffic for the round's profiles/
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles
rm -rf $out; mkdir -p $out
cd $root
python bench.py > $out/bench_hex100.json 2>$out/bench_hex100.err
python bench.py --workload hex100c > $out/bench_hex100c.json 2>>$out/bench_hex100.err
python bench.py --workload hex100L --steps 50 --warmup 5 > $out/bench_hex100L.json 2>>$out/bench_hex100.err
python bench.py --workload hex215 --no-cpu-baseline --steps 50 --warmup 5 > $out/bench_hex215.json 2>>$out/bench_hex100.err
python bench.py --workload hex300 --no-cpu-baseline --steps 20 --warmup 2 > $out/bench_hex300.json 2>>$out/bench_hex100.err
python bench.py --workload cavity215 --no-cpu-baseline --steps 50 --warmup 5 > $out/bench_cavity215.json 2>>$out/bench_hex100.err
python bench.py --workload cavity215c --no-cpu-baseline --steps 20 --warmup 2 > $out/bench_cavity215c.json 2>>$out/bench_hex100.err
python bench.py --workload cavity100c --steps 20 --warmup 2 > $out/bench_cavity100c.json 2>>$out/bench_hex100.err
python bench.py --workload hex100B --steps 50 --warmup 5 > $out/bench_hex100B.json 2>>$out/bench_hex100.err
python bench.py --workload cavity100B --no-cpu-baseline --steps 50 --warmup 5 > $out/bench_cavity100B.json 2>>$out/bench_hex100.err
python bench.py --workload hex215B --no-cpu-baseline --steps 30 --warmup 3 > $out/bench_hex215B.json 2>>$out/bench_hex100.err
python bench.py --workload hex100cB --no-cpu-baseline --steps 50 --warmup 5 > $out/bench_hex100cB.json 2>>$out/bench_hex100.err
cd /tmp && export TMPDIR=/tmp
for wl in hex100 hex100c hex100B; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_$wl -- python3 $root/bench.py --no-cpu-baseline --workload $wl > /dev/null 2>&1
  cp $out/rocprof_$wl/*/*kernel_stats.csv $out/rocprof_${wl}_kernel_stats.csv
  rm -rf $out/rocprof_$wl
done
cd $root
bash scripts/measure_traffic.sh hex100 10 > /dev/null
bash scripts/measure_traffic.sh hex215 5 > /dev/null
cp gpurun_out/traffic_hex100.json gpurun_out/traffic_hex215.json $out/
ls -la $out
