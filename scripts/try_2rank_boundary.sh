#!/bin/bash
# two ranks on the single GPU of a gpurun box through the gloo debug backend: DistributedSmoother + boundary point smoothing
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 SMOOTHMESH_SHARE_GPU=1 SMOOTHMESH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/check_dist_boundary.py > gpurun_out/dist_bnd.log 2>&1
echo "== check_dist_boundary exit $?"; grep "^rank" gpurun_out/dist_bnd.log; grep -i "error\|Traceback" gpurun_out/dist_bnd.log | head -5
for wl in hex40B hex40cB; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 10 --warmup 2 --workload $wl --no-cpu-baseline > gpurun_out/try2_$wl.log 2>&1
  echo "== $wl exit $?"; grep "^{" gpurun_out/try2_$wl.log | python scripts/bench_summary.py | head -1; grep -i "error\|Traceback" gpurun_out/try2_$wl.log | head -3
done
