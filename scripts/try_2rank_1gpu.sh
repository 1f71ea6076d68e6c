#!/bin/bash
# experiment: two ranks on the single GPU of a gpurun box (RCCL normally rejects duplicate GPUs)
export SMOOTHMESH_BENCH_FULL=1   # bench.py prints its full document (not the compact driver line) on stdout
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export SMOOTHMESH_SHARE_GPU=1
export SMOOTHMESH_BACKEND=gloo
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 2 --workload hex40 > gpurun_out/try2.log 2>&1; grep -v "^[W|^W1|^***" gpurun_out/try2.log | head -60
