#!/bin/bash
# two ranks on the single GPU of a gpurun box through the gloo debug backend: hex, hex + layers, polyhedral
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 SMOOTHMESH_SHARE_GPU=1 SMOOTHMESH_BACKEND=gloo
for wl in hex40 hex40L cavity40c cavity40L; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 10 --warmup 2 --workload $wl > gpurun_out/try2_$wl.log 2>&1
  echo "== $wl exit $?"; grep "^{" gpurun_out/try2_$wl.log | python scripts/bench_summary.py | head -1; grep -i "error\|Traceback" gpurun_out/try2_$wl.log | head -3
done
