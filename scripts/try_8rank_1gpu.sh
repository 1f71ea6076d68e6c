#!/bin/bash
# debugging aid: the 8-rank (2x2x2) driver path on the single GPU of a gpurun box, gloo instead of RCCL
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 SMOOTHMESH_SHARE_GPU=1 SMOOTHMESH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 10 --warmup 2 --workload ${1:-hex24} > gpurun_out/try8.log 2>&1
grep "^{" gpurun_out/try8.log | python scripts/bench_summary.py | head -8; grep -i "error\|Traceback" gpurun_out/try8.log | head -5
