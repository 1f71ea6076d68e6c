#!/bin/bash
# The first run on a node with >= 2 MI355X: everything that has never executed on more than one device, in the order in which
# a failure is cheapest to understand.  Every step runs in a FRESH process (a process that has touched the GPU is never re-used
# or re-exec'd), every step exits non-zero on a mismatch, and the script stops at the first failure.
#   usage: scripts/first_multi_gpu.sh [nGPUs]      (default: all visible; run from the repository root)
# Writes one log per step under gpurun_out/first_multi_gpu/.
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
N=${1:-$(python3 -c 'import torch; print(torch.cuda.device_count())')}
OUT=gpurun_out/first_multi_gpu
mkdir -p "$OUT"
if [ "$N" -lt 2 ]; then echo "needs >= 2 GPUs (found $N)"; exit 2; fi
step=0
run() {   # run <name> <command...>
    step=$((step + 1))
    local name=$1; shift
    local log="$OUT/$(printf %02d $step)_$name.log"
    echo "== step $step: $name"
    echo "   $*"
    if "$@" > "$log" 2>&1; then echo "   ok  ($log)"; else echo "   FAILED (exit $?) -- see $log"; tail -20 "$log"; exit 1; fi
}

# 0. the build is the one in the tree, the single-GPU path still stands
run build        python3 -c 'import __graft_entry__ as g; g.build()'
run smoke        python3 -c 'import __graft_entry__ as g; g.smoke()'

# 1. the three RCCL tests that are skipped on a 1-GPU box (grouped ncclSend / ncclRecv between real devices: Python driver on
#    2 and on min(N, 8) ranks, the C++ front-end's -parallel on 2 ranks)
run rccl_python_2      python3 -m pytest -x -q -m gpu "tests/test_gpu_multirank.py::test_distributed_smoother_polyhedral_over_rccl[2]"
if [ "$N" -ge 8 ]; then run rccl_python_8 python3 -m pytest -x -q -m gpu "tests/test_gpu_multirank.py::test_distributed_smoother_polyhedral_over_rccl[8]"; fi
run rccl_cli_2         python3 -m pytest -x -q -m gpu "tests/test_gpu_cli.py::test_parallel_case_over_rccl_when_the_box_has_two_gpus"

# 1b. the same two-rank case with the one-kernel-per-step form of the iteration (the A/B of the multi-role launches: step 1 ran
#     k_geom_halo / k_smooth_halo in order and -- overlap True -- flagged, the log lines say which) and, traced: one profiler per
#     rank started by the launcher (which never touches the GPU; the interpreter directly behind `--`), so that the first run also
#     yields the exchange timeline (kernel trace + statistics per rank under $OUT/trace/)
run rccl_one_kernel_per_step env SMGPU_HALO_MERGED=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 scripts/check_dist_poly.py
run rccl_traced        sh -c "cd /tmp && TMPDIR=/tmp python3 -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29553 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$OUT/trace -- python3 $PWD/scripts/check_dist_poly.py"

# 1c. the C++ front-end with its exchanges on a stream of their own (flagged arrangement; opt-in there until this step has passed)
run rccl_cli_2_exchange_stream env SMOOTHMESH_EXCHANGE_STREAM=1 python3 -m pytest -x -q -m gpu "tests/test_gpu_cli.py::test_parallel_case_over_rccl_when_the_box_has_two_gpus"

# 2. irregular sub-domains over RCCL (a rank without shared points, ragged counts), one process per device
run rccl_irregular     env CHECK_IRREGULAR=two_blocks:41 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29551 scripts/check_dist_poly.py

# 3. the peer-store transport between two DEVICES (so far both processes sat behind one L2): without and with the full
#    system-scope fences.  It stays opt-in until this step has passed with SMGPU_PUSH_FENCE=0.
for fence in 0 1; do
    run push_fence$fence env SMOOTHMESH_EXCHANGE=push SMGPU_PUSH_FENCE=$fence python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2956$fence scripts/check_dist_poly.py
    run push_bnd_fence$fence env SMOOTHMESH_EXCHANGE=push SMGPU_PUSH_FENCE=$fence python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2957$fence scripts/check_dist_boundary.py
done

# 4. the bench lines the driver will ask for (each starts its own ranks; parity_check must be ok in every line)
for g in 2 4 8; do
    [ "$g" -le "$N" ] || continue
    run bench_$g sh -c "python3 bench.py --gpus $g > $OUT/bench_$g.json && python3 scripts/check_bench_line.py $OUT/bench_$g.json $g"
done
echo "all $step steps passed"
