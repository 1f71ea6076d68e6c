#!/bin/bash
# run on the GPU box: benchmark lines + rocprofv3 kernel statistics + PMC traffic / SQ counters for the round's profiles/
# usage: scripts/collect_profiles.sh [round tag, default r6]   -> gpurun_out/profiles_<tag>/ (copy what is to be judged to profiles/<tag>/)
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
tag=${1:-r6}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
rm -rf $out; mkdir -p $out
cd $root
B="timeout 300 python bench.py"
timeout 1500 python bench.py > $out/bench_default.json 2>$out/bench.err                                   # hex100 + configs[] (hex100c, cavity215, cavity215c) + cpu_baseline
$B --workload hex100c --no-configs > $out/bench_hex100c.json 2>>$out/bench.err
$B --workload hex215 --no-cpu-baseline --no-configs --steps 50 --warmup 5 > $out/bench_hex215.json 2>>$out/bench.err
$B --workload hex300 --no-cpu-baseline --no-configs --steps 20 --warmup 2 > $out/bench_hex300.json 2>>$out/bench.err
$B --workload cavity215 --no-cpu-baseline --no-configs --steps 50 --warmup 5 > $out/bench_cavity215.json 2>>$out/bench.err
$B --workload cavity215c --no-cpu-baseline --no-configs --steps 200 --warmup 5 > $out/bench_cavity215c.json 2>>$out/bench.err   # configs[3] as written: 200 iterations
$B --workload cavity100c --no-configs --steps 40 --warmup 5 > $out/bench_cavity100c.json 2>>$out/bench.err
$B --workload hex100L --no-configs --steps 50 --warmup 5 > $out/bench_hex100L.json 2>>$out/bench.err
$B --workload hex100B --no-configs --steps 50 --warmup 5 > $out/bench_hex100B.json 2>>$out/bench.err
SMGPU_WALK=host $B --workload cavity215c --no-cpu-baseline --no-configs --steps 30 --warmup 5 > $out/bench_cavity215c_hostwalk.json 2>>$out/bench.err  # round-1 replay place, A/B
cd /tmp && export TMPDIR=/tmp
for wl in hex100 hex100c cavity215 cavity215c; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_$wl -o p -- python3 $root/bench.py --no-cpu-baseline --no-configs --workload $wl --steps 50 --warmup 5 > /dev/null 2>&1
  cp $out/rocprof_$wl/*/p_kernel_stats.csv $out/rocprof_${wl}_kernel_stats.csv 2>/dev/null
  cp $out/rocprof_$wl/p_kernel_stats.csv $out/rocprof_${wl}_kernel_stats.csv 2>/dev/null
  rm -rf $out/rocprof_$wl
done
cd $root
export SMGPU_SIDE_STREAM=0   # counter collection serialises kernels: no cross-stream waits
for wl in hex100 hex215 cavity215 cavity215c; do
  BENCH_EXTRA="--no-configs" timeout 600 bash scripts/measure_traffic.sh $wl 10 > /dev/null 2>&1
  cp gpurun_out/traffic_$wl.json $out/ 2>/dev/null
done
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d $out/sq$i -- python3 $root/bench.py --workload hex100 --steps 20 --warmup 2 --no-cpu-baseline --no-configs > /dev/null 2>&1
done
cd $root
python3 scripts/pmc_summary.py $out > $out/pmc_sq_hex100.txt 2>/dev/null
rm -rf $out/sq1 $out/sq2 $out/sq3
# what every kernel executes per launch, by instruction type (SQ_INSTS_VALU_* etc.)
for wl in hex100 cavity215 cavity215c; do
  timeout 900 bash scripts/pmc_inst_mix.sh $wl 10 > /dev/null 2>&1
  cp gpurun_out/inst_mix_$wl/summary.txt $out/inst_mix_$wl.txt 2>/dev/null; rm -rf gpurun_out/inst_mix_$wl
done
SMGPU_VERBOSE=2 timeout 300 python3 scripts/create_time.py cavity215 > $out/setup_phases_cavity215_final.txt 2>&1
# one rank of eight: both transports (RCCL send / recv groups; peer stores with a self-mapping), with kernel timelines
timeout 900 bash scripts/probe_timeline.sh > /dev/null 2>&1
cp gpurun_out/probe_timeline/probe.txt $out/probe_rank_of_8_rccl.txt; cp gpurun_out/probe_timeline/timeline.txt $out/probe_rank_of_8_timeline_rccl.txt
cp gpurun_out/probe_timeline/timeline_boundary.txt $out/probe_rank_of_8_timeline_boundary_rccl.txt
SMOOTHMESH_EXCHANGE=push timeout 900 bash scripts/probe_timeline.sh > /dev/null 2>&1
cp gpurun_out/probe_timeline/probe.txt $out/probe_rank_of_8_push.txt; cp gpurun_out/probe_timeline/timeline.txt $out/probe_rank_of_8_timeline_push.txt
cp gpurun_out/probe_timeline/timeline_boundary.txt $out/probe_rank_of_8_timeline_boundary_push.txt
# the same probe with the absolute timeline of all streams (in order / exchange stream), the arrangement equality check, and the
# set-up phases on this box's host
timeout 600 bash scripts/probe_timeline2.sh > /dev/null 2>&1
cp gpurun_out/probe_timeline2/inorder.txt $out/probe_rank_of_8_abs_timeline_inorder_rccl.txt; cp gpurun_out/probe_timeline2/overlap.txt $out/probe_rank_of_8_abs_timeline_flagged_rccl.txt
(timeout 300 python3 scripts/check_arrangements.py 2>&1 | grep -E "reference|same bits|DIFFERENT|arrangements"; SMOOTHMESH_EXCHANGE=push timeout 300 python3 scripts/check_arrangements.py 2>&1 | grep -E "reference|same bits|DIFFERENT|arrangements") > $out/check_arrangements.txt
SMGPU_HALO_MERGED=0 timeout 300 python3 scripts/probe_rank_of_8.py 2>&1 | grep -E "rank 0 of 8|inorder|overlap|serial|transport" > $out/probe_rank_of_8_rccl_one_kernel_per_step.txt
SMGPU_HALO_MERGED=0 SMOOTHMESH_EXCHANGE=push timeout 300 python3 scripts/probe_rank_of_8.py 2>&1 | grep -E "rank 0 of 8|inorder|overlap|serial|transport" > $out/probe_rank_of_8_push_one_kernel_per_step.txt
# round 6: configs[4]'s rank (the 430^3-base polyhedral mesh's box 0, constraints on) and the tool a user runs, end to end
timeout 900 python3 scripts/probe_rank_of_8.py cavity215c 2>&1 | grep -v "^\[W\|Warning" > $out/probe_rank_of_8_cavity215c_rccl.txt
timeout 900 python3 scripts/cli_clocktime.py 215 --serial-io > $out/cli_cavity215c_clocktime.txt 2>&1
ls -la $out
